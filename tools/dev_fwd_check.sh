#!/bin/bash
# Development loop: decoder parity subsets, then the bench line (no extras) A/B over one switch ($1=VAR), B=1 replay times.
set -u
VAR=${1:-WMD_DEFER_REDUCE}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse.py tests/test_gpu_configs.py -q -x -p no:cacheprovider -k "decoder or config2 or config4 or config3 or replay or bound or sparse" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in 1 0 1 0; do env $VAR=$v python bench.py --no-train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$v', d['value'], d['ms_per_step'], r['heads_and_idwt'], r['kernels_ms_per_step'].get('conv_splitk_reduce_kernel'), r['kernels_ms_per_step'].get('head_chain_kernel'))"; done
for v in 1 0; do env $VAR=$v python tools/probes/sparse_host_cost.py 1 2>&1 | grep "host"; done

#!/bin/bash
# Development loop for trunk-kernel changes: conv parity subset, the benchmarked-mode test, then the bench line (no extras), A/B.
set -u
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -q -x -p no:cacheprovider -k "conv or decoder or config2 or dgrad or wino" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in 1 0 1 0; do WMD_X4=$v python bench.py --no-train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('X4=$v', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['all_conv_kernels']['ms_per_step'], {k:v for k,v in r['kernels_ms_per_step'].items() if 'wino' in k})"; done

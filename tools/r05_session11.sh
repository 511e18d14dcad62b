#!/bin/bash
# round 5, session 11: chained completion of levels 4..2 -- parity, bench A/B, one-frame dense timeline
OUT=$PWD/gpurun_out/r05_s11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "chained_completion or dense_decoder or config2 or golden" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for v in 1 0 1 0; do
  WMD_SHIFTSUM_CHAIN=$v timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_chain$v.json 2>> $OUT/bench.err
  python -c "import json; d=json.load(open('$OUT/bench_chain$v.json')); print('chain=$v', d['value'], d['ms_per_step'], d['ms_per_step_p10_median_p90'], d['roofline']['heads_and_idwt'], {k:v for k,v in d['roofline']['kernels_ms_per_step'].items() if 'head' in k})"
done
for v in 1 0; do WMD_SHIFTSUM_CHAIN=$v timeout 600 python tools/config_bench.py sparse 2>&1 | grep -E "dense decoder batch 1"; done

#!/bin/bash
# A/B: L14 on <8,64,4,8> (tuner's choice) vs <8,32,2,8> (round 5's) inside the replayed step
for i in 1 2 3 4; do
  for c in default l14_8x32 step_tuned; do
    if [ $c = default ]; then unset WMD_TUNE_CACHE; else export WMD_TUNE_CACHE=/tmp/ab_$c.json; cp tools/ab/$c.json /tmp/ab_$c.json; fi
    python bench.py --steps 50 --warmup 10 --no-train --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$c', d['ms_per_step'], d['ms_per_step_p10_median_p90'], r['kernel'], r['avg_launch_us'], r['frac'], r['all_conv_kernels']['ms_per_step'], r['wino32_family']['frac'])
"
  done
done

#!/bin/bash
# round 5, session 7: what do the instructions between the MFMAs cost?  quarter-position kernel with the transform arithmetic (16),
# the weight-fragment reads (32), the patch reads (64) compiled out at run time (results wrong by construction), DMA off (12) throughout
OUT=$PWD/gpurun_out/r05_s7; mkdir -p $OUT
ST=$PWD/tools/probes/_build/libwmd_stamps.so
for m in 12 28 44 76 124; do
  echo "=== WMD_DBG_MODE=$m" >> $OUT/issue_experiment.txt
  WMD_DBG_MODE=$m WMD_LIB_PATH=$ST timeout 600 python tools/wino32_microbench.py 14 10 13 --ksplits 1 --iters 10 --no-old --cfgs "8,16,8;4,32,8" 2>&1 | grep -E "== layer|ks1" | sed 's/max rel err.*//' >> $OUT/issue_experiment.txt
done
cat $OUT/issue_experiment.txt

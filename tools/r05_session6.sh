#!/bin/bash
# round 5, session 6: how much of a launch is LDS-DMA traffic?  Timing experiments (results wrong by construction): weight pieces
# skipped after the first two chunks (4), patch pieces too (12), patch only (8) -- half-position and quarter-position kernels
OUT=$PWD/gpurun_out/r05_s6; mkdir -p $OUT
ST=$PWD/tools/probes/_build/libwmd_stamps.so
for m in 0 4 8 12; do
  echo "=== WMD_DBG_MODE=$m" >> $OUT/dma_experiment.txt
  WMD_DBG_MODE=$m WMD_LIB_PATH=$ST timeout 600 python tools/wino32_microbench.py 14 10 9 13 --ksplits 1 --iters 10 --no-old --cfgs "8,16,8;4,32,8;8,32,2,8;8,64,4,8" 2>&1 | grep -E "== layer|ks1" | sed 's/max rel err.*//' >> $OUT/dma_experiment.txt
done
cat $OUT/dma_experiment.txt

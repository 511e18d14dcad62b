#!/bin/bash
OUT=$PWD/gpurun_out/r05_s10; mkdir -p $OUT
for lf in 800 200 0; do
  WMD_SPARSE_LIST_FROM=$lf timeout 900 python tools/config_bench.py sparse > $OUT/sparse_lf$lf.txt 2>&1
  echo "== WMD_SPARSE_LIST_FROM=$lf"; grep -E "dense decoder batch 1, hip|contour masks, densities|thresh  0.15|thresh  0.20|thresh  0.05|injected density 0.10" $OUT/sparse_lf$lf.txt
done

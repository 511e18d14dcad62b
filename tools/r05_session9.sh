#!/bin/bash
# round 5, session 9: MASKED / LIST instantiations of conv_wino32q_kernel -- parity (every masked configuration, the sparse decoders in
# every form, the LIST convolution tests) and the sparse workloads A/B (WMD_LIST_FAMILY=17: the half-position LIST kernel)
OUT=$PWD/gpurun_out/r05_s9; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "block_sparse or sparse or config4 or poisoned" > $OUT/pytest_sparse.log 2>&1; tail -4 $OUT/pytest_sparse.log
for fam in 18 17; do
  WMD_LIST_FAMILY=$fam timeout 900 python tools/config_bench.py sparse sparse-throughput > $OUT/sparse_workloads_fam$fam.txt 2>&1
  echo "== LIST family $fam"; grep -E "dense decoder batch 1, hip|contour masks, densities 0.10 0.03|thresh  0.15|thresh  0.20|one batch of 12|contour masks of density|density 1.00|density 0.10 \(mean|thresh 0.20" $OUT/sparse_workloads_fam$fam.txt
done

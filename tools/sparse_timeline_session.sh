#!/bin/bash
# Timelines (rocprofv3 kernel trace of one hipGraph replay) of the sparse / dense KITTI decoder at batch 1 and 12 (development aid).
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/tl
tl() { name=$1; shift; rm -rf /tmp/tlx; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tlx -- python $R/tools/sparse_timeline.py run "$@" > /tmp/o.txt 2>&1); python $R/tools/sparse_timeline.py parse /tmp/tlx/*/*kernel_trace.csv > $R/gpurun_out/tl/$name.txt; echo "$name: $(tail -1 $R/gpurun_out/tl/$name.txt)"; }
tl dense_b1 dense
WMD_SPARSE_TILES=0 tl sparse_b1_gather 0.15
tl sparse_b1_lists 0.15
WMD_SPARSE_LISTS=0 tl sparse_b1_r03form 0.15
tl sparse_b1_lists_contour 0.05 contour=0.1,0.03,0.01
tl dense_b12 dense batch=12
tl sparse_b12_d0.1 0.05 batch=12 density=0.1
tl sparse_b12_thr0.2 0.2 batch=12
tl sparse_b12_contour 0.05 batch=12 contour=0.1,0.03,0.01
WMD_SPARSE_LISTS=0 tl sparse_b12_contour_r03form 0.05 batch=12 contour=0.1,0.03,0.01

#!/bin/bash
# Round-4 sparse session on the GPU box: parity of the work-list form, then the workloads A/B against the round-3 tile form
# (WMD_SPARSE_LISTS=0) on the same box, then replay timelines.   usage: bash tools/r04_sparse_session.sh <tag> [tests] [bench] [ab] [tl]
set -u
TAG=${1:-r04s}; shift
WHAT=${*:-tests bench ab tl}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cp profiles/r03_tune_cache.json $OUT/tune_cache.json
export WMD_TUNE_CACHE=$OUT/tune_cache.json
for w in $WHAT; do
case $w in
tests)
    timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_configs.py -q -x -k "sparse or config4" -p no:cacheprovider > $OUT/tests.txt 2>&1
    tail -n 15 $OUT/tests.txt ;;
bench)
    timeout 600 python tools/config_bench.py sparse sparse-throughput > $OUT/sparse_workloads.txt 2>&1
    grep "sparse\|throughput\|Error\|error" $OUT/sparse_workloads.txt | tail -n 45 ;;
ab)
    WMD_SPARSE_LISTS=0 timeout 600 python tools/config_bench.py sparse sparse-throughput > $OUT/sparse_workloads_r03form.txt 2>&1
    grep "sparse\|throughput\|Error\|error" $OUT/sparse_workloads_r03form.txt | tail -n 45 ;;
tl)
    R=$PWD; mkdir -p gpurun_out/tl
    tl() { name=$1; shift; rm -rf /tmp/tlx; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlx -- python $R/tools/sparse_timeline.py run "$@" > /tmp/o.txt 2>&1); python $R/tools/sparse_timeline.py parse /tmp/tlx/*/*kernel_trace.csv > $OUT/tl_$name.txt; echo "$name: $(tail -1 $OUT/tl_$name.txt)"; }
    tl dense_b1 dense
    tl sparse_b1_thr0.15 0.15
    tl sparse_b1_contour 0.05 contour=0.1,0.03,0.01
    tl sparse_b12_contour 0.05 batch=12 contour=0.1,0.03,0.01
    tl sparse_b12_d0.1 0.05 batch=12 density=0.1
    for f in dense_b1 sparse_b1_thr0.15 sparse_b1_contour sparse_b12_contour sparse_b12_d0.1; do echo "==== $f"; cat $OUT/tl_$f.txt; done > $OUT/timelines.txt ;;
esac
done
ls $OUT

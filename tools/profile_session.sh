#!/bin/bash
# One profiling session on the GPU box (run from the repo root through gpurun); raw outputs go to gpurun_out/<tag>/,
# tools/make_profile_summary.py turns them into the committed profiles/ artefacts.
#   usage: bash tools/profile_session.sh <tag> [fwd] [bwd] [sparse] [tests]        (default: all four)
set -u
TAG=${1:-session}; shift
WHAT=${*:-fwd bwd sparse tests}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
pmc_passes() {   # $1 = name prefix, rest = command: FETCH_SIZE / WRITE_SIZE / SQ counters in separate passes (never with a trace domain)
    local name=$1; shift
    local SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
    (cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${name}_fetch -- "$@" > $OUT/pmc_${name}_fetch.log 2>&1)
    python $REPO/tools/pmc_reduce.py $OUT/pmc_${name}_fetch
    (cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${name}_write -- "$@" > $OUT/pmc_${name}_write.log 2>&1)
    python $REPO/tools/pmc_reduce.py $OUT/pmc_${name}_write
    (cd /tmp && timeout 400 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_${name}_mfma -- "$@" > $OUT/pmc_${name}_mfma.log 2>&1)
    python $REPO/tools/pmc_reduce.py $OUT/pmc_${name}_mfma
}
for w in $WHAT; do
case $w in
retune)   # regenerate the committed tile / split-K choices (every workload bench.py touches): profiles/<tag>_tune_cache.json
    WMD_BENCH_RETUNE=1 WMD_TUNE_CACHE=$OUT/tune_cache.json timeout 1500 python bench.py --no-cpu-baseline > $OUT/bench_retune.json 2> $OUT/bench_retune.err
    tail -c 300 $OUT/bench_retune.json; echo; python -c "import json; print(len(json.load(open('$OUT/tune_cache.json'))), 'tuned keys')" ;;
fwd)
    [ -f $OUT/tune_cache.json ] && cp $OUT/tune_cache.json profiles/${TAG}_tune_cache.json     # this session's retune, if any
    python bench.py > $OUT/bench.json 2> $OUT/bench.err
    # the trunk layers' problem signatures with the kernel + grid each ran on (joins the counters to layers, not to grid sizes)
    WMD_CONV_VERBOSE=1 WMD_BENCH_GRAPH=0 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train 2>&1 >/dev/null | grep "sig conv" | sort -u > $OUT/conv_plan.txt
    tail -c 400 $OUT/bench.json; echo
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --no-cpu-baseline --no-train > $OUT/stats.log 2>&1)
    cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/stats
    WMD_BENCH_GRAPH=0 pmc_passes fwd python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train
    # north_star's second resolution (KITTI ResNet50 1024x320, batch 8): plan dump + the same three counter passes
    WMD_CONV_VERBOSE=1 WMD_BENCH_GRAPH=0 python bench.py --workload fwd-1024 --steps 1 2>&1 >/dev/null | grep "sig conv" | sort -u > $OUT/conv_plan_1024.txt
    WMD_BENCH_GRAPH=0 pmc_passes fwd1024 python $REPO/bench.py --workload fwd-1024 --steps 3 ;;
bwd)   # WMD_TUNE_CACHE: the first run tunes and stores its choices, the counter passes replay them (no tuning launches in the counters)
    export WMD_TUNE_CACHE=$OUT/train_tune_cache.json
    python tools/train_profile.py > $OUT/train_profile_r18_640x192_bs12.txt 2>&1
    python tools/train_profile.py --chans 64,256,512,1024,2048 --height 320 --width 1024 --batch 8 > $OUT/train_profile_r50_1024x320_bs8.txt 2>&1
    python tools/train_profile.py --nyu > $OUT/train_profile_nyu_densenet161_640x480_bs4.txt 2>&1
    head -4 $OUT/train_profile_*.txt
    pmc_passes bwd python $REPO/tools/train_profile.py
    unset WMD_TUNE_CACHE ;;
sparse)
    python tools/config_bench.py sparse sparse-throughput > $OUT/sparse_workloads.txt 2>&1
    WMD_SPARSE_LISTS=0 python tools/config_bench.py sparse sparse-throughput > $OUT/sparse_workloads_r03form.txt 2>&1    # round-3 tile form, same box
    grep "sparse\|throughput" $OUT/sparse_workloads.txt | tail -n 50
    bash tools/sparse_timeline_session.sh > $OUT/sparse_timelines.txt 2>&1      # one graph replay each, kernel by kernel
    for f in dense_b1 sparse_b1_gather sparse_b1_lists sparse_b1_r03form sparse_b1_lists_contour dense_b12 sparse_b12_d0.1 sparse_b12_thr0.2 sparse_b12_contour sparse_b12_contour_r03form; do
        echo "==== $f" >> $OUT/sparse_timelines.txt; cat gpurun_out/tl/$f.txt >> $OUT/sparse_timelines.txt; done
    export WMD_TUNE_CACHE=$OUT/sparse_tune_cache.json
    python tools/sparse_profile.py 0.15 > $OUT/sparse_profile_thr0.15.txt 2>&1
    pmc_passes sparse python $REPO/tools/sparse_profile.py 0.15
    unset WMD_TUNE_CACHE ;;
tests)
    timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > $OUT/gpu_tests.txt 2>&1
    python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/gpu_tests.txt 2>&1
    grep -n "passed\|failed\|smoke" $OUT/gpu_tests.txt | tail -n 3 ;;
esac
done
du -sh $OUT; ls $OUT

#!/bin/bash
# One profiling session on the GPU box (run from the repo root through gpurun); raw outputs go to gpurun_out/<tag>/,
# tools/make_profile_summary.py turns them into the committed profiles/ artefacts.
#   usage: bash tools/profile_session.sh <tag>
set -u
TAG=${1:-session}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 1. re-tune tile/split-K choices for config 2 and run the full bench line (with the CPU baseline)
WMD_BENCH_RETUNE=1 WMD_TUNE_CACHE=$OUT/tune_cache.json python bench.py > $OUT/bench_retune.json 2> $OUT/bench_retune.err
export WMD_TUNE_CACHE=$OUT/tune_cache.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
# 2. kernel trace of the same command (no CPU leg)
#    (two-stream replay off: co-running kernels stretch each other's durations, the summary is compared with bench.py's
#     serial per-launch hipEvents; the overlapped trace is kept next to it)
(cd /tmp && WMD_TWO_STREAM_GRAPHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $OLDPWD/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_two_stream -- python $OLDPWD/bench.py --no-cpu-baseline > $OUT/stats_two_stream.log 2>&1)
# 3. PMC passes, separate runs, eager launches so every kernel is a dispatch of its own
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass
    name=$1; shift
    (cd /tmp && WMD_BENCH_GRAPH=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1)
done
ls $OUT

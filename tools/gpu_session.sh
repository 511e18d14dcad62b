#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): GPU test-suite, the bench line, SQ counter passes over the
# eager bench (what bounds the Winograd trunk kernels).  Raw outputs -> gpurun_out/<tag>/.
#   usage: bash tools/gpu_session.sh <tag> [tests|bench|pmc|traffic ...]
set -u
TAG=${1:-session}; shift
WHAT=${*:-tests bench pmc}
OUT=$PWD/gpurun_out/$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
for w in $WHAT; do
case $w in
tests)
    timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1
    tail -n 25 $OUT/pytest_gpu.log ;;
bench)
    [ -f $OUT/tune_cache.json ] && export WMD_TUNE_CACHE=$OUT/tune_cache.json
    timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
    tail -c 1500 $OUT/bench.json; tail -n 5 $OUT/bench.err ;;
retune)
    WMD_BENCH_RETUNE=1 WMD_TUNE_CACHE=$OUT/tune_cache.json timeout 900 python bench.py > $OUT/bench_retune.json 2> $OUT/bench_retune.err
    tail -c 600 $OUT/bench_retune.json ;;
stats)
    [ -f $OUT/tune_cache.json ] && export WMD_TUNE_CACHE=$OUT/tune_cache.json
    (cd /tmp && WMD_TWO_STREAM_GRAPHS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --no-cpu-baseline --no-train > $OUT/stats.log 2>&1) ;;
pmc)
    [ -f $OUT/tune_cache.json ] && export WMD_TUNE_CACHE=$OUT/tune_cache.json
    (cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1)
    python - "$OUT/counters_list.txt" > $OUT/pmc_passes.txt <<'PY'
import sys
txt = open(sys.argv[1]).read()
passes = [
    ("time", "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"),
    ("insts", "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32"),
    ("lds", "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"),
    ("grbm", "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F32"),
]
for name, cs in passes:
    ok = [c for c in cs.split() if c in txt]
    print(name, " ".join(ok))
PY
    cat $OUT/pmc_passes.txt
    while read name counters; do
        [ -z "$counters" ] && continue
        (cd /tmp && WMD_BENCH_GRAPH=0 timeout 300 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$name -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train > $OUT/pmc_$name.log 2>&1)
        python tools/pmc_summary.py "$OUT/pmc_$name/*/*counter_collection.csv" conv_wino > $OUT/pmc_${name}_wino.txt 2>&1
    done < $OUT/pmc_passes.txt
    head -n 40 $OUT/pmc_time_wino.txt ;;
traffic)
    [ -f $OUT/tune_cache.json ] && export WMD_TUNE_CACHE=$OUT/tune_cache.json
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
        set -- $pass
        name=$1; shift
        (cd /tmp && WMD_BENCH_GRAPH=0 timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train > $OUT/pmc_$name.log 2>&1)
    done ;;
esac
done
ls $OUT

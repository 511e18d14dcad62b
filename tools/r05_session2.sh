#!/bin/bash
# round 5, session 2: direct gather offsets + line-coalesced output stores in conv_wino32_kernel -- stamps per variant, parity, A/B bench
OUT=$PWD/gpurun_out/r05_s2; mkdir -p $OUT
ST=$PWD/tools/probes/_build/libwmd_hip_stamps.so
J="14:8,32,2,8:1 10:8,32,2,8:1 9:8,32,2,8:1 13:8,32,2,8:1 1:6,40,2,8:4 6:8,16,1,8:1"
for v in "0 0 0" "1 0 0" "0 1 0" "1 1 0" "0 0 1" "0 0 3"; do
  set -- $v
  STAMPS_TAG="direct=$1 coalesce=$2 dbg=$3" WMD_W32_DIRECT=$1 WMD_W32_COALESCE=$2 WMD_DBG_MODE=$3 WMD_LIB_PATH=$ST timeout 300 python tools/probes/stamps_probe.py $J >> $OUT/stamps.txt 2>&1
done
timeout 900 python tools/wino32_microbench.py 14 10 6 9 1 --ksplits 1,2 --iters 8 --no-old > $OUT/micro_new.txt 2>&1
WMD_W32_DIRECT=0 WMD_W32_COALESCE=0 timeout 900 python tools/wino32_microbench.py 14 10 6 9 1 --ksplits 1,2 --iters 8 --no-old > $OUT/micro_old.txt 2>&1
for i in 1 2; do
WMD_W32_DIRECT=0 WMD_W32_COALESCE=0 timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_old$i.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_new$i.json 2>> $OUT/bench.err
done
WMD_W32_DIRECT=1 WMD_W32_COALESCE=0 timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_direct_only.json 2>> $OUT/bench.err
WMD_W32_DIRECT=0 WMD_W32_COALESCE=1 timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_coalesce_only.json 2>> $OUT/bench.err
for f in $OUT/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['ms_per_step_p10_median_p90'], d['roofline']['kernels_ms_per_step'])"; done
grep MISMATCH $OUT/micro_new.txt | head

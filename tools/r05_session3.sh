#!/bin/bash
# round 5, session 3: s_setprio in prologue / epilogue, 16-channel chunks, DMA issue window (SP) variants, poisoned-pool test
OUT=$PWD/gpurun_out/r05_s3; mkdir -p $OUT
ST=$PWD/tools/probes/_build/libwmd_stamps.so
J="14:8,32,2,8:1 10:8,32,2,8:1 9:8,32,2,8:1 13:8,32,2,8:1 1:6,40,2,8:4"
for p in 0 1 2 3; do
  STAMPS_TAG="prio=$p" WMD_W32_PRIO=$p WMD_LIB_PATH=$ST timeout 300 python tools/probes/stamps_probe.py $J >> $OUT/stamps_prio.txt 2>&1
done
STAMPS_TAG="ck16" WMD_LIB_PATH=$ST timeout 300 python tools/probes/stamps_probe.py 9:8,32,2,16:1 5:6,40,2,16:2 1:6,40,2,16:1 >> $OUT/stamps_ck16.txt 2>&1
timeout 900 python tools/wino32_microbench.py 9 5 1 0 13 --ksplits 1,2,4 --iters 8 --no-old --cfgs "8,32,2,16;6,40,2,16;8,32,2,8;6,40,2,8" > $OUT/micro_ck16.txt 2>&1
for v in hip sp13 sp12; do
  L=$PWD/tools/probes/_build/libwmd_$v.so; [ $v = hip ] && L=$PWD/wavelet_monodepth_amd/libwmd_hip.so
  [ -f $L ] || continue
  WMD_LIB_PATH=$L timeout 600 python tools/wino32_microbench.py 14 10 6 9 1 --ksplits 1,4 --iters 8 --no-old --cfgs "8,32,2,8;6,40,2,8;8,16,1,8" > $OUT/micro_$v.txt 2>&1
  WMD_LIB_PATH=$L timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_$v.json 2>> $OUT/bench.err
done
for p in 1 2 3; do
  WMD_W32_PRIO=$p timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_prio$p.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench_hip2.json 2>> $OUT/bench.err
for f in $OUT/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['ms_per_step_p10_median_p90'], d['roofline']['kernels_ms_per_step'])"; done
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "poisoned or config4 or sparse" > $OUT/pytest_sparse.log 2>&1; tail -5 $OUT/pytest_sparse.log

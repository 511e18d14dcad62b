#!/bin/bash
# round 5, session 1: where a conv_wino32 block's cycles go (stamps), the small layers over every configuration, baseline bench
OUT=$PWD/gpurun_out/r05_s1; mkdir -p $OUT
WMD_LIB_PATH=$PWD/tools/probes/_build/libwmd_hip_stamps.so timeout 300 python tools/probes/stamps_probe.py > $OUT/stamps.txt 2>&1
STAMPS_BATCH=1 WMD_LIB_PATH=$PWD/tools/probes/_build/libwmd_hip_stamps.so timeout 300 python tools/probes/stamps_probe.py 14:8,32,2,8:1 10:8,32,2,8:1 > $OUT/stamps_b1.txt 2>&1
timeout 600 python tools/wino32_microbench.py 0 1 5 9 13 --ksplits 1,2,4,8 --iters 10 > $OUT/micro_small.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-train > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
head -50 $OUT/stamps.txt

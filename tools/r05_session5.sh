#!/bin/bash
# round 5, session 5: conv_wino32q_kernel (quarter-position waves, three blocks per CU): parity of the forced configurations, per-layer A/B
OUT=$PWD/gpurun_out/r05_s5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "every_tile_configuration or winograd_configurations or block_sparse_every" > $OUT/pytest_cfg.log 2>&1; tail -6 $OUT/pytest_cfg.log
timeout 900 python tools/wino32_microbench.py 14 10 6 9 13 --ksplits 1,2 --iters 10 --no-old --cfgs "8,16,8;4,32,8;8,32,2,8;8,16,1,8;16,16,2,8;8,64,4,8;4,64,2,8" > $OUT/micro_q.txt 2>&1
grep -E "== layer|ks1" $OUT/micro_q.txt

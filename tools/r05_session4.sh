#!/bin/bash
# round 5, session 4: training forward of the heads on the fused kernels -- parity, then the decoder fwd+bwd profile A/B
OUT=$PWD/gpurun_out/r05_s4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_graph.py -q -p no:cacheprovider -x > $OUT/pytest_bwd.log 2>&1; tail -15 $OUT/pytest_bwd.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider -x -k "grad or train or config3 or config5 or config2" > $OUT/pytest_cfg.log 2>&1; tail -5 $OUT/pytest_cfg.log
export WMD_TUNE_CACHE=$OUT/tune.json; cp profiles/r04_tune_cache.json $WMD_TUNE_CACHE
for v in 1 0; do
  WMD_TRAIN_FUSED_HEADS=$v timeout 600 python tools/train_profile.py > $OUT/train_r18_fused$v.txt 2>&1
  WMD_TRAIN_FUSED_HEADS=$v timeout 600 python tools/train_profile.py --chans 64,256,512,1024,2048 --height 320 --width 1024 --batch 8 > $OUT/train_r50_fused$v.txt 2>&1
done
head -12 $OUT/train_r18_fused1.txt; head -5 $OUT/train_r18_fused0.txt; head -3 $OUT/train_r50_fused1.txt; head -3 $OUT/train_r50_fused0.txt

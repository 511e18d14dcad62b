#!/bin/bash
# run on the GPU box: time the ablated builds (tools/w32_ablate.sh) on two layers
OUT=gpurun_out/$1; mkdir -p $OUT
CFGS="8,32,2,8;16,32,4,8;8,16,1,8"
echo "== full" > $OUT/ablate.txt
python tools/wino32_microbench.py 14 10 --cfgs "$CFGS" --ksplits 1 --no-old >> $OUT/ablate.txt 2>&1
for d in 1 2 3 4 8 16 31; do
  echo "== W32_DBG=$d" >> $OUT/ablate.txt
  WMD_LIB_PATH=$PWD/build_abl/libwmd_dbg$d.so python tools/wino32_microbench.py 14 10 --cfgs "$CFGS" --ksplits 1 --no-old 2>&1 | grep -v "^/opt" >> $OUT/ablate.txt
done
cat $OUT/ablate.txt

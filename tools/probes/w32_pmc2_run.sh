#!/bin/bash
# instruction-cache / fetch counters of conv_wino32_kernel (and the 16x16x4 kernel beside it) on layer 14
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1)
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" $OUT/counters_list.txt | sort -u | tr '\n' ' ' > $OUT/sqc_names.txt
cat $OUT/sqc_names.txt; echo
i=0
for counters in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmcb$i -- python $REPO/tools/wino32_microbench.py 14 --cfgs "8,32,2,8" --ksplits 1 --iters 4 > $OUT/pmcb$i.log 2>&1)
  python tools/pmc_summary.py "$OUT/pmcb$i/*/*counter_collection.csv" conv_wino > $OUT/pmcb${i}_summary.txt 2>&1
  grep -A12 "wino32\|8, 32, 1, 2, 4, 8" $OUT/pmcb${i}_summary.txt | head -60
done

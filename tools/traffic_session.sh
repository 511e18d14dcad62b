#!/bin/bash
# tools/traffic_probe.py under rocprofv3 --pmc, one TCC counter group per pass (4 TCC slots), per (layer, split form)
# -> gpurun_out/traffic/<layer>_<ks>_<group>.csv (reduced by tools/pmc_reduce.py)
set -u
OUT=$PWD/gpurun_out/traffic; REPO=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
declare -A G
G[rd]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum"
G[wr]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_ATOMIC_sum"
G[hm]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_NORMAL_WRITEBACK_sum"
G[ev]="TCC_NORMAL_EVICT_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_READ_sum TCC_WRITE_sum"
for spec in ${*:-L1:-4 L1:4 L1:1 L0:-4 L0:4 L0:1 L14:1}; do
  layer=${spec%%:*}; ks=${spec##*:}
  for g in rd wr hm ev; do
    d=$OUT/${layer}_${ks}_$g
    (cd /tmp && timeout 200 rocprofv3 --pmc ${G[$g]} --output-format csv -d $d -- python $REPO/tools/traffic_probe.py $layer $ks > $d.log 2>&1)
    python $REPO/tools/pmc_reduce.py $d > /dev/null
  done
  tail -1 $OUT/${layer}_${ks}_rd.log
done
ls $OUT | head -50

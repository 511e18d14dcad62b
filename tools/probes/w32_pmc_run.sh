#!/bin/bash
# run on the GPU box: SQ counter passes over conv_wino32_kernel on layer 14 (tools/wino32_microbench.py)
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT; REPO=$PWD
CFG=${2:-16,32,4,8}
export TMPDIR=/tmp
i=0
for counters in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
                "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32" \
                "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
                "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc$i -- python $REPO/tools/wino32_microbench.py 14 --cfgs "$CFG" --ksplits 1 --no-old --iters 5 > $OUT/pmc$i.log 2>&1)
  python tools/pmc_summary.py "$OUT/pmc$i/*/*counter_collection.csv" conv_wino32 > $OUT/pmc${i}_summary.txt 2>&1
  cat $OUT/pmc${i}_summary.txt
done

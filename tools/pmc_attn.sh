#!/bin/bash
# PMC passes on the attention micro-benchmark: tools/pmc_attn.sh "<B N H>"
export TMPDIR=/tmp
O=gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS"; do
  rocprofv3 --pmc $set -d $O/p$i -o p -- python tools/attn_micro.py $1 3 > /dev/null 2>&1
  python tools/pmc_summary.py $(find $O/p$i -name "*_results.db" | head -1) 2>/dev/null | grep attn | cut -c1-30,70-140
  i=$((i+1))
done
rm -rf $O

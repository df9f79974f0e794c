#!/bin/bash
# PMC passes on the conv micro-benchmark: tools/pmc_conv.sh "<N H W C K ks stride which>"
export TMPDIR=/tmp
O=gpurun_out/pmc_conv
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR"; do
  rocprofv3 --pmc $set -d $O/p$i -o p -- python tools/conv_micro.py $1 3 > /dev/null 2>&1
  python tools/pmc_summary.py $(find $O/p$i -name "*_results.db" | head -1) 2>/dev/null | grep -i "conv\|wgrad" | cut -c1-40,70-140
  i=$((i+1))
done
rm -rf $O

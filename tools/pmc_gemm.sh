#!/bin/bash
# PMC passes on the gemm micro-benchmark: tools/pmc_gemm.sh "<M N K epi>"  (separate passes, no tracing domains)
export TMPDIR=/tmp
A="$1"
O=gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $set -d $O/p$i -o p -- python tools/gemm_micro.py $A 5 > /dev/null 2>&1
  python tools/pmc_summary.py $(find $O/p$i -name "*_results.db" | head -1) 2>/dev/null | grep gemm
  i=$((i+1))
done

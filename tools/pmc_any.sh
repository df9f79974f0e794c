#!/bin/bash
# PMC passes over any command: tools/pmc_any.sh <kernel-name filter> -- <command ...>   (counters in their own runs, no trace domains)
export TMPDIR=/tmp
F="$1"; shift; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_o
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_o -o p -- "$@" > /dev/null 2>&1)
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for fn in glob.glob('/tmp/pmc_o/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if "$F" not in r['Kernel_Name']: continue
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:28s} per-dispatch {v / max(n,1):.4g}  (n={n})")
PY
done

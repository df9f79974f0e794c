#!/bin/bash
# usage: tools/prof_conv.sh <tag> "<shape args>" ... ; true kernel durations via rocprofv3 kernel-trace
export TMPDIR=/tmp
tag=$1; shift
i=0
for a in "$@"; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/pc_${tag}_$i -o p -- python tools/conv_micro.py $a 20 > /dev/null 2>&1
  python - <<PY
import sqlite3,re
c = sqlite3.connect('gpurun_out/pc_${tag}_$i/p_results.db')
for r in c.execute("select name,total_calls,average from top_kernels where name like '%conv%' and name not like '%weight%'"):
    print("$a".ljust(34), re.sub(r'\(anonymous namespace\)::|void |\(.*$','',r[0])[:58].ljust(58), "%7.1f us" % r[2])
PY
  i=$((i+1))
done

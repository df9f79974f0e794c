#!/bin/bash
# usage: tools/prof_env.sh "<ENV=..>" "<shape args>" : prints true kernel duration
export TMPDIR=/tmp
env $1 rocprofv3 --kernel-trace --stats -d gpurun_out/pe_tmp -o p -- python tools/conv_micro.py $2 20 > /dev/null 2>&1
python - <<PY
import sqlite3,re
c = sqlite3.connect('gpurun_out/pe_tmp/p_results.db')
for r in c.execute("select name,total_calls,average from top_kernels where name like '%conv%' and name not like '%weight%'"):
    print("$1".ljust(44), "$2".ljust(30), "%7.1f us" % r[2])
PY
rm -rf gpurun_out/pe_tmp

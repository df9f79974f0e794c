#!/bin/bash
# usage: tools/prof_all.sh "<ENV=..>" "<shape args>" : prints true durations of every kernel in the run
export TMPDIR=/tmp
env $1 rocprofv3 --kernel-trace --stats -d gpurun_out/pe_tmp -o p -- python tools/conv_micro.py $2 20 > /dev/null 2>&1
python - <<PY
import sqlite3,re
c = sqlite3.connect('gpurun_out/pe_tmp/p_results.db')
out=[]
for r in c.execute("select name,total_calls,average from top_kernels where (name like '%conv%' or name like '%wgrad%') and name not like '%weight%'"):
    out.append("%s %.1f us" % (re.sub(r'\(anonymous namespace\)::|void |\(.*$|<.*$','',r[0]), r[2]))
print("$1".ljust(26), "$2".ljust(30), " | ".join(out))
PY
rm -rf gpurun_out/pe_tmp

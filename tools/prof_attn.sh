#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/pa; rocprofv3 --kernel-trace --stats -d gpurun_out/pa -o p -- python tools/attn_micro.py $1 10 > /dev/null 2>&1
python - <<PY
import sqlite3,re,glob
c = sqlite3.connect(glob.glob('gpurun_out/pa/**/*_results.db', recursive=True)[0])
for r in c.execute("select name,total_calls,average from top_kernels where name like '%attn%'"):
    print("$1".ljust(16), re.sub(r'\(anonymous namespace\)::|void |\(.*$','',r[0])[:40].ljust(40), "%7.1f us" % r[2])
PY
rm -rf gpurun_out/pa

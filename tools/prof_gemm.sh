#!/bin/bash
# usage: tools/prof_gemm.sh "<M N K epi>" ... : rocprof kernel-trace durations of the gemm kernel per shape
export TMPDIR=/tmp
i=0
for a in "$@"; do
  rm -rf gpurun_out/pg_$i
  rocprofv3 --kernel-trace --stats -d gpurun_out/pg_$i -o p -- python tools/gemm_micro.py $a 20 > /dev/null 2>&1
  python - <<PY
import sqlite3,re,glob
c = sqlite3.connect(glob.glob('gpurun_out/pg_$i/**/*_results.db', recursive=True)[0])
M,N,K,epi = map(int, "$a".split()[:4])
for r in c.execute("select name,total_calls,average from top_kernels where name like '%gemm%'"):
    print("$a".ljust(24), re.sub(r'\(anonymous namespace\)::|void |\(.*$','',r[0])[:50].ljust(50), "%7.1f us  %6.0f TFLOP/s" % (r[2], 2.0*M*N*K/r[2]/1e6))
PY
  rm -rf gpurun_out/pg_$i
  i=$((i+1))
done

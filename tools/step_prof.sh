#!/bin/bash
# per-symbol in-step kernel times of one bench.py configuration: tools/step_prof.sh "<ENV=..>" [bench args]  -> top $TOP (24) symbols, with their time per step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
E="$1"; shift
cd /tmp; rm -rf /tmp/sp_o
env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_o -o p -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" > /tmp/sp_o.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/sp_o.log
python3 - <<PY
import csv, glob, re, os
TOP = int(os.environ.get('TOP', '24'))
for fn in glob.glob('/tmp/sp_o/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"  total kernel time {tot/1e6:.1f} ms")
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:TOP]:
        n = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Name'])
        print(f"  {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %  {float(r['TotalDurationNs'])/1e3/110:7.1f} us/step")
PY

#!/bin/bash
# per-kernel durations of the wgrad_bench cases: tools/wgrad_prof.sh "<ENV=..>" <case filter>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/wp_o
env $1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp_o -o p -- $R/tools/ubench/wgrad_bench "$2" 20 > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
for fn in glob.glob('/tmp/wp_o/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Name'])
        print(f"  $1 $2 {n[:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY

#!/bin/bash
# per-kernel, per-shape durations of the BatchNorm launches of tools/bn_bench.py from a rocprofv3 kernel trace:  tools/bn_prof.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/bp_o
rocprofv3 --kernel-trace --output-format csv -d /tmp/bp_o -o p -- python $R/tools/bn_bench.py 20 > /tmp/bp_o.log 2>&1
python3 - <<PY
import csv, glob, re, collections
for fn in glob.glob('/tmp/bp_o/**/*kernel_trace.csv', recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Kernel_Name'])
        if 'bn_' not in n: continue
        d[(n, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    for (n, g), v in sorted(d.items()):
        v.sort()
        print(f"{n[:58]:58s} grid {g:>8s} calls {len(v):4d} median {v[len(v)//2]:6.1f} us  min {v[0]:6.1f}")
PY

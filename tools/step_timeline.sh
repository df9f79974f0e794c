#!/bin/bash
# timeline of ONE training step of bench.py from a rocprofv3 kernel trace: tools/step_timeline.sh "<ENV=..>" [bench args]
# prints every dispatch between two consecutive sgd_kernel launches: start (us from the step's first kernel), duration, queue, name
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
E="$1"; shift
cd /tmp; rm -rf /tmp/tl_o
env $E rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_o -o p -- python $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" > /tmp/tl_o.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/tl_o.log
python3 - <<PY
import csv, glob, re
for fn in glob.glob('/tmp/tl_o/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    name = lambda r: re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Kernel_Name'])
    sg = [i for i, r in enumerate(rows) if name(r).startswith('sgd_')]
    far = [i for i in range(len(sg) - 1) if sg[i + 1] - sg[i] > 10]      # (one or two optimizer launches per step)
    a, b = sg[far[len(far) // 2]], sg[far[len(far) // 2] + 1]
    t0 = int(rows[a]['End_Timestamp'])
    qs = {}
    busy_end = 0
    for r in rows[a + 1:b + 1]:
        q = qs.setdefault(r['Queue_Id'], len(qs))
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        print(f"{s:9.1f} {e - s:7.1f} q{q} {'    ' * q}{name(r)[:70]}")
PY

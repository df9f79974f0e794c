#!/bin/bash
# HBM-side traffic of the BatchNorm launches of tools/bn_bench.py: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace
# domains), per kernel symbol and grid size.  Bytes = 2 * FETCH_SIZE KB (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KB.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bnp_$c
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/bnp_$c -o p -- python $R/tools/bn_bench.py 10 > /dev/null 2>&1)
done
python3 - <<PY
import csv, glob, re, collections
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(f'/tmp/bnp_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] != c: continue
            n = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r['Kernel_Name'])
            if 'bn_' not in n: continue
            a = agg[(n, r.get('Grid_Size', r.get('Grid_Size_X', '')))]; a[0] += 1; a[1] += float(r['Counter_Value'])
    val[c] = {k: v[1] / v[0] for k, v in agg.items()}
print(f"{'kernel':58s} {'grid':>8s} {'read MB':>9s} {'write MB':>9s}   (per dispatch; the 262144-thread grids are the 32x32 stage: 33.5 MB per tensor;")
print(f"{'':58s} {'':>8s} {'':>9s} {'':>9s}    131072-thread grids mix the 16x16 / 8x8 / 4x4 stages: 16.8 / 8.4 / 4.2 MB per tensor)")
for k in sorted(val["FETCH_SIZE"]):
    rd = 2 * val["FETCH_SIZE"][k] * 1024 / 1e6
    wr = val["WRITE_SIZE"].get(k, 0.0) * 1024 / 1e6
    print(f"{k[0][:58]:58s} {k[1]:>8s} {rd:9.1f} {wr:9.1f}")
PY

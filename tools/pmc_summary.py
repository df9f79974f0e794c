"""sum PMC counters per kernel from a rocprofv3 rocpd sqlite db: python tools/pmc_summary.py <db>"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = list(c.execute("select * from counters_collection"))
ix = {n: i for i, n in enumerate(cols)}
agg = {}
for r in rows:
    k = (re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', r[ix['kernel_name']] if 'kernel_name' in ix else str(r[ix.get('name', 0)]))[:70], r[ix['counter_name']])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += r[ix['value']]
for (kn, cn), (n, v) in sorted(agg.items()):
    print(f"{kn:70s} {cn:16s} n={n:5d} sum={v:.4g} avg={v/n:.4g}")

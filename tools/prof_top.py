"""print the rocprofv3 (rocpd sqlite) top-kernel table: python tools/prof_top.py <results.db> [n]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':80s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>8s} {'%':>6s}")
for r in rows[:n]:
    nm = re.sub(r'\(anonymous namespace\)::|void ', '', r[0])
    nm = re.sub(r'\(.*$', '', nm)
    print(f"{nm[:80]:80s} {r[1]:6d} {r[2]/1e3:10.2f} {r[3]:8.1f} {r[4]:6.2f}")
print(f"total kernel time {tot/1e3:.2f} ms over the whole run")

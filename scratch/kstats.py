"""Per-kernel duration summary from a rocprofv3 rocpd database: python scratch/kstats.py <db> [filter]"""
import sqlite3, sys, glob, re
db = sys.argv[1]
if not db.endswith('.db'):
    db = glob.glob(db + '/*/*.db')[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, (end-start) from kernels").fetchall()
agg = {}
for n, d in rows:
    if flt and flt not in n: continue
    n = re.sub(r'\(.*', '', n)
    a = agg.setdefault(n, [0, 0, 1e18, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{a[0]:6d} avg {a[1]/a[0]/1000:8.2f} us  min {a[2]/1000:7.2f}  max {a[3]/1000:7.2f}  {100*a[1]/tot:5.1f}%  {n[:110]}")
print("total ms", tot / 1e6)

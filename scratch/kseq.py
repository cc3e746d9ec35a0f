"""Kernel sequence of ONE replayed step from a rocprofv3 rocpd database: python scratch/kseq.py <dir>"""
import sqlite3, sys, glob, re
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = 'kernels' if 'kernels' in tabs else [t for t in tabs if 'kernel_dispatch' in t][0]
rows = c.execute(f"select name, start, end, grid_x, workgroup_x from {kt} order by start").fetchall() if kt == 'kernels' else None
names = [re.sub(r'\(.*', '', r[0]) for r in rows]
# one step = from one fill_f32/set_int to the next; take the last complete one
idx = [i for i, n in enumerate(names) if 'set_int_kernel' in n]
a, b = idx[-3], idx[-2]
# extend backwards to the forward's start (stem_fwd)
st = max(i for i in range(a) if 'stem_fwd' in names[i]) if any('stem_fwd' in n for n in names[:a]) else a
prev_end = None
for i in range(st, st + (b - a)):
    n, s, e, gx, wx = rows[i]
    gap = (s - prev_end) / 1000 if prev_end else 0
    print(f"{(e - s) / 1000:8.2f} us  gap {gap:6.2f}  wg {gx // max(wx,1):6d}  {names[i][:90]}")
    prev_end = e

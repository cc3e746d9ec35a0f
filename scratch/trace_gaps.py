"""Per-step kernel time and idle gaps from a rocprofv3 kernel trace CSV: python scratch/trace_gaps.py <kernel_trace.csv> [marker]
A step is cut at every launch of `marker` (default: stem_fwd_kernel); the last 10 whole steps are averaged."""
import csv, re, sys
from collections import defaultdict
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_fwd_kernel"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])))
rows.sort()
cuts = [i for i, r in enumerate(rows) if marker in r[2]]
steps = list(zip(cuts[:-1], cuts[1:]))[-11:-1]
tot = busy = 0.0
per = defaultdict(lambda: [0, 0.0])
gaps = defaultdict(lambda: [0, 0.0])
for a, b in steps:
    seg = rows[a:b]
    tot += (rows[b][0] - seg[0][0]) / 1e3
    end = seg[0][0]
    for s, e, n in seg:
        busy += (e - max(s, end)) / 1e3 if e > end else 0.0
        if s > end:
            g = gaps[n]; g[0] += 1; g[1] += (s - end) / 1e3
        end = max(end, e)
        p = per[n]; p[0] += 1; p[1] += (e - s) / 1e3
    if rows[b][0] > end:
        g = gaps["<next step>"]; g[0] += 1; g[1] += (rows[b][0] - end) / 1e3
k = len(steps)
print(f"{k} steps: {tot / k:.1f} us per step, device busy {busy / k:.1f} us, idle {(tot - busy) / k:.1f} us, {sum(p[0] for p in per.values()) / k:.0f} launches")
print("kernel time per step (top 25):")
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / k:9.1f} us  x{c / k:6.1f}  {n[:100]}")
print("idle time in FRONT of (top 15):")
for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t / k:9.1f} us  x{c / k:6.1f}  avg {t / max(c, 1):6.2f}  {n[:100]}")

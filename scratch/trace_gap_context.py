"""Which launch precedes every device-idle gap > 2 us in a rocprofv3 kernel trace: python scratch/trace_gap_context.py <kernel_trace.csv>"""
import csv, re, sys
from collections import Counter
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:70]))
rows.sort()
rows = rows[len(rows) // 2:]          # the timed half of the run
c = Counter(); tot = Counter()
end = rows[0][1]; prev = rows[0][2]
for s, e, n in rows[1:]:
    if s - end > 2000:
        c[(prev, n)] += 1; tot[(prev, n)] += (s - end) / 1e3
    if e > end:
        end = e; prev = n
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:15]:
    print(f"{v / c[k]:8.2f} us x{c[k]:4d}   after {k[0]}\n                      before {k[1]}")

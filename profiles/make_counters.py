#!/usr/bin/env python3
"""Per-class averages of arbitrary rocprofv3 --pmc counters (one pass): python profiles/make_counters.py <dir> <out.json>
Classes as in make_traffic.py.  Values are per launch, summed over the chip as rocprofv3 reports them."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic import classify

f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    c = classify(r["Kernel_Name"])
    if c:
        acc[c][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[c][r["Counter_Name"]] += 1
out = {c: {k: acc[c][k] / cnt[c][k] for k in acc[c]} | {"launches_sampled": max(cnt[c].values())} for c in acc}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for c, v in sorted(out.items()):
    print(c, {k: (round(x, 1) if isinstance(x, float) else x) for k, x in v.items()})

"""Condense rocprofv3 output (kernel stats CSV + PMC counter CSVs) into a short text + JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


summary = {}
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats:", os.path.relpath(f, out))
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        print("%-90s calls=%s avg_ns=%s total_ns=%s pct=%s" % (name, r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
        summary.setdefault("kernel_stats", []).append({"name": r.get("Name"), "calls": int(r.get("Calls", 0)), "avg_ns": float(r.get("AverageNs", 0)), "pct": float(r.get("Percentage", 0))})
for tag in ("fetch", "write"):
    files = find("pmc_%s/**/*counter_collection.csv" % tag)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:90], r.get("Counter_Name"))
            agg[k][0] += float(r.get("Counter_Value", 0))
            agg[k][1] += 1
    print("== PMC", tag)
    for (kn, cn), (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print("%-90s %s per_launch=%.1f (n=%d)" % (kn, cn, v / max(c, 1), c))
        summary.setdefault("pmc", []).append({"kernel": kn, "counter": cn, "per_launch": v / max(c, 1), "launches": c})
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)

"""Per-launch means of the PMC counters of one kernel from a rocprofv3 counter_collection.csv."""
import csv
import json
import sys
from collections import defaultdict

path, kern = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))     # counter -> dispatch -> value (summed over XCDs / dimensions)
for row in csv.DictReader(open(path)):
    if kern not in row.get("Kernel_Name", ""):
        continue
    acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = {}
for c, d in acc.items():
    vals = list(d.values())
    out[c] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals), "min": min(vals), "max": max(vals),
              "per_launch": [d[k] for k in sorted(d, key=lambda x: int(x))][:64]}      # (in dispatch order)
print(json.dumps(out, indent=1))

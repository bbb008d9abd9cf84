"""Summarise a rocprofv3 counter_collection.csv: per-kernel sum of each counter over dispatches matching a name."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
nd = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "")
        if pat not in k:
            continue
        short = k.split("(")[0][-60:]
        acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
        nd[short].add(row.get("Dispatch_Id"))
for k, d in acc.items():
    n = max(1, len(nd[k]))
    print(f"kernel {k}  dispatches={n}")
    for c, v in sorted(d.items()):
        print(f"  {c:32s} total={v:.6g}  per_dispatch={v / n:.6g}")

"""Per-launch-shape durations from a rocprofv3 --kernel-trace CSV: python tools/experiments/kernel_shapes.py <kernel_trace.csv> [name filter]
Groups the dispatches of each kernel by grid size and prints count / mean / min microseconds, in order of first appearance."""
import csv
import sys
from collections import OrderedDict

path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
groups = OrderedDict()
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        key = (name.split("(")[0][-48:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, gx, gy, gz), d in groups.items():
    print(f"{name:48s} grid {gx:>8s} x {gy:>6s} x {gz:>4s}  n={len(d):4d}  mean {sum(d) / len(d):8.1f} us  min {min(d):8.1f} us")

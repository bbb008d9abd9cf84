"""Kernel sequence of the last full iteration from a rocprofv3 --kernel-trace CSV: python tools/experiments/iter_sequence.py <kernel_trace.csv> <marker>
One line per kernel between the last two launches of the marker kernel: start offset, duration, idle gap in front (device-wide), queue, name."""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[3]]
a, b = marks[-2], marks[-1]
win = rows[a + 1:b + 1]
t0 = win[0][0]
cur = win[0][0]
gap_total = 0
for s, e, q, n in win:
    gap = max(0, s - cur)
    gap_total += gap
    print(f"+{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  q{q:>2s}  {n.split('(')[0][-70:]}")
    cur = max(cur, e)
print(f"{len(win)} kernels, wall {(cur - t0) / 1e3:.1f} us, device idle {gap_total / 1e3:.1f} us")

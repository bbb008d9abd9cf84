"""Timeline of one iteration from a rocprofv3 --kernel-trace CSV: python tools/experiments/timeline.py <kernel_trace.csv> <marker kernel substring>
Takes the window between the last two launches of the marker kernel (e.g. k_adam = one training iteration), merges runs of
kernels per queue into segments and prints, per segment: queue, start offset, busy time, number of kernels, leading kernel name;
plus the idle gaps of the whole device (no kernel running on any queue) longer than 20 us."""
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
print(f"window: {len(win)} kernels, {(win[-1][1] - t0) / 1e6:.2f} ms wall, {sum(e - s for s, e, _, _ in win) / 1e6:.2f} ms summed kernel time")
short = lambda n: n.split("(")[0].split("::")[-1][:40]
# per-queue segments: consecutive kernels of a queue with gaps < 30 us
segs = {}
for s, e, q, n in win:
    L = segs.setdefault(q, [])
    if L and s - L[-1][1] < 30000:
        L[-1][1] = max(L[-1][1], e); L[-1][2] += e - s; L[-1][3] += 1
    else:
        L.append([s, e, e - s, 1, short(n)])
allseg = sorted((sg[0], q, sg) for q, L in segs.items() for sg in L)
for s, q, (ss, ee, busy, cnt, name) in allseg:
    if ee - ss > 100000:
        print(f"  queue {q:>3s}  +{(ss - t0) / 1e6:7.2f} ms  span {(ee - ss) / 1e6:6.2f} ms  busy {busy / 1e6:6.2f} ms  {cnt:4d} kernels  first: {name}")
# device idle gaps
ev = sorted((s, e) for s, e, _, _ in win)
cur = ev[0][1]
for s, e in ev[1:]:
    if s - cur > 20000:
        print(f"  idle gap {(s - cur) / 1e3:7.1f} us at +{(cur - t0) / 1e6:.2f} ms")
    cur = max(cur, e)

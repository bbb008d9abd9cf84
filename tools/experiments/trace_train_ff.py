"""Phase trace of the training step's fused feed-forward kernels (k_ff<false>, k_ff<true>, k_ff_wgrad): builds libdfx with
-DDFX_TRACE_FF (stamps of wave 0 of every 29th workgroup: tag + shader clock + HW_ID), runs tools/bench_train.py once and prints the
average duration of each phase in shader cycles.  GPU box only; rebuilds the plain library afterwards.

    python tools/experiments/trace_train_ff.py [out.txt]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from difffacto_amd import build  # noqa: E402

NAMES = {0: "k_ff<false>", 1: "k_ff<true>", 2: "k_ff_wgrad (producer wave 0)", 3: "k_ff_wgrad (consumer wave 4)"}


def analyse(path, out):
    rows = collections.defaultdict(list)
    for line in open(path):
        m = re.match(r"kernel (\d+) wg (\d+):(.*)", line)
        if not m:
            continue
        ev = [(int(a), int(h, 16), int(c)) for a, h, c in re.findall(r"(\d+)@([0-9a-f]+):(\d+)", m.group(3))]
        rows[int(m.group(1))].append((int(m.group(2)), ev))
    for k in sorted(rows):
        print(f"== {NAMES[k]}: {len(rows[k])} traced workgroups (wave 0), shader cycles", file=out)
        tot = collections.defaultdict(list)
        first = None
        for wg, ev in rows[k]:
            t = {}
            seq = []
            prev = ev[0][2]
            for tag, hw, c in ev:
                d = (c - prev) & 0xffffffffff
                seq.append((tag, d))
                prev = c
            life = (ev[-1][2] - ev[0][2]) & 0xffffffffff
            tot["lifetime"].append(life)
            # phases
            idx = {tag: i for i, (tag, _, _) in enumerate(ev)}   # last occurrence
            clk = lambda tag: ev[idx[tag]][2] if tag in idx else None
            if k < 2:
                def span(a, b, name):
                    if clk(a) is not None and clk(b) is not None:
                        tot[name].append((clk(b) - clk(a)) & 0xffffffffff)
                first_tag = {tag: i for i, (tag, _, _) in reversed(list(enumerate(ev)))}
                t1, t2, t3 = ev[first_tag[1]][2], ev[first_tag[2]][2], ev[first_tag[3]][2]
                tot["prologue (rows, LayerNorm, frag stores)"].append(t2 - t1)
                tot["chunk loop (16 chunks)"].append(t3 - t2)
                # inside the loop: sum by segment type
                seg = collections.defaultdict(int)
                p = t2
                for tag, hw, c in ev:
                    if c <= t2 or c > t3 or tag < 10:
                        continue
                    seg[tag] += c - p
                    p = c
                for tag, v in seg.items():
                    tot[f"  loop seg ending at tag {tag}"].append(v)
                if k == 1:
                    span(3, 4, "epilogue a: gradient lo halves in, xhat3 / dh into the accumulator layout, LN3 sums")
                    span(4, 5, "epilogue b: LN3 backward, 4 column sums (d b2)")
                    span(5, 6, "epilogue c: read hin, LN2, sim (8 MFMA), xhat2")
                    span(6, 7, "epilogue d: softmax, dP (8 MFMA), dxn2 (8 MFMA), LN2 backward, stores, 3 x 4 column sums")
                    span(7, 8, "epilogue e: workgroup reduction + last stores landed")
                else:
                    span(3, 9, "epilogue: store h2")
            elif k == 3:
                arr = [(tag, c) for tag, hw, c in ev]
                for a_, b_, name in ((20, 21, "consumer: wait at arrive()"), (21, 22, "consumer: turn tile k around (4 MFMAs, packs, LDS writes)"),
                                     (22, 23, "consumer: reads + 24 MFMAs of tile k - 1 issued"), (23, 20, "consumer: loop back")):
                    v = [y[1] - x[1] for x, y in zip(arr, arr[1:]) if x[0] == a_ and y[0] == b_]
                    if v:
                        tot[name].append(sum(v) / len(v))
            else:
                arr = [(tag, c) for tag, hw, c in ev]
                for a_, b_, name in ((11, 14, "producer: tile reads + 24 MFMAs issued"), (14, 12, "producer: MFMA results + GEGLU arithmetic"),
                                     (12, 13, "producer: 6 fragment stores to LDS")):
                    v = [y[1] - x[1] for x, y in zip(arr, arr[1:]) if x[0] == a_ and y[0] == b_]
                    if v:
                        tot[name].append(sum(v) / len(v))
                waits = [b[1] - a[1] for a, b in zip(arr, arr[1:]) if a[0] == 10 and b[0] == 11]
                work = [b[1] - a[1] for a, b in zip(arr, arr[1:]) if a[0] == 11 and b[0] == 10]
                if waits:
                    tot["per tile: wait at arrive() (vmcnt + barrier)"].append(sum(waits) / len(waits))
                if work:
                    tot["per tile: producer work between barriers"].append(sum(work) / len(work))
            if first is None:
                first = (wg, seq)
        for name, v in tot.items():
            print(f"   {name:95s} mean {sum(v) / len(v):10.0f}   min {min(v):9.0f}   max {max(v):9.0f}", file=out)
        cus = collections.Counter((ev[0][1] >> 4) & 0xfff for _, ev in rows[k])
        print(f"   distinct (SE, CU, SIMD... HW_ID>>4) values among the traced workgroups: {len(cus)}", file=out)
        if first:
            print(f"   workgroup {first[0]} raw (tag, delta): {first[1][:70]}", file=out)
        if k < 2:   # first wait of the prologue (tag 1 -> 16, when stamped) and lifetime by workgroup id: later rounds of workgroups start out of step
            rowsw = []
            for wg, ev in rows[k]:
                ft = {tag: c for tag, hw, c in reversed(ev)}
                if 16 in ft and 1 in ft:
                    rows_only = f"({(ft[17] - ft[1]) & 0xffffffffff} rows)" if 17 in ft else ""
                    rowsw.append(f"{wg}:{(ft[16] - ft[1]) & 0xffffffffff}{rows_only}/{(ev[-1][2] - ev[0][2]) & 0xffffffffff}")
            if rowsw:
                print("   workgroup id : first wait / lifetime   " + "  ".join(rowsw), file=out)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04", "trace_train_ff.txt")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE_FF"])
    raw = "/tmp/ff_trace_raw.txt"
    env = dict(os.environ, DFX_TRACE_FF_OUT=raw)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py")], env=env, capture_output=True, text=True)
        with open(out_path, "w") as f:
            print("# " + (r.stdout.strip().splitlines() or ["(no output)"])[-1], file=f)
            if r.returncode != 0:
                print(r.stderr[-2000:], file=f)
            else:
                analyse(raw, f)
    finally:
        build.build(force=True, verbose=False)
    print(open(out_path).read())


if __name__ == "__main__":
    main()

"""Phase trace of the chained training forward (k_ff_fwd_chain): wave 0 of every 29th workgroup stamps (tag, shader clock) through all five blocks
(-DDFX_TRACE_FF build; the later blocks append to the first block's row).  Prints, per block, the mean cycles of
    top (1) -> first wait + barrier (16) -> attention sub-block (17) -> its barrier (18) -> LN3 / fragment stores / b2 (2) -> 16 chunks (13 compute, 14 counted wait,
    15 barrier) -> loop end (3) -> row stores (9 / next block's 1)
GPU box only; rebuilds the plain library afterwards.     python tools/experiments/trace_ff_chain.py [out.txt] [bench_train args]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from difffacto_amd import build  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "trace_ff_chain.txt")
    extra = [a for a in sys.argv[2:] if a.startswith("-D")]
    bargs = [a for a in sys.argv[2:] if not a.startswith("-D")]
    build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE_FF"] + extra)
    raw = "/tmp/ff_trace_raw.txt"
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py")] + bargs, env=dict(os.environ, DFX_TRACE_FF_OUT=raw), capture_output=True, text=True)
    finally:
        build.build(force=True, verbose=False)
    lines = ["# " + (r.stdout.strip().splitlines() or ["(no output)"])[-1][:170] + "   [flags: " + " ".join(extra) + "]"]
    if r.returncode != 0:
        lines.append(r.stderr[-2000:])
    else:
        per = collections.defaultdict(lambda: collections.defaultdict(list))   # block -> phase -> samples
        life = []
        for line in open(raw):
            m = re.match(r"kernel 0 wg (\d+):(.*)", line)
            if not m:
                continue
            ev = [(int(a), int(c)) for a, h, c in re.findall(r"(\d+)@([0-9a-f]+):(\d+)", m.group(2))]
            blk, prev, seg = -1, None, collections.defaultdict(int)
            for tag, c in ev:
                if tag == 1:
                    if blk >= 0:
                        for k, v in seg.items():
                            per[blk][k].append(v)
                        per[blk]["9 -> next block's 1"].append(c - prev)
                    blk += 1
                    seg = collections.defaultdict(int)
                else:
                    seg[{16: "a 1 -> 16 first wait + barrier", 17: "b attention sub-block (LN2, 16 MFMA, softmax, swaps)", 18: "c barrier behind it", 2: "d LN3, fragment stores, b2",
                         13: "e loop: compute (16 chunks)", 14: "f loop: counted wait", 15: "g loop: barrier", 3: "h loop exit", 9: "i row stores / head"}.get(tag, str(tag))] += c - prev
                prev = c
            for k, v in seg.items():
                per[blk][k].append(v)
            life.append(ev[-1][1] - ev[0][1])
        lines.append(f"k_ff_fwd_chain: {len(life)} traced workgroups (wave 0), lifetime mean {sum(life) / max(1, len(life)):.0f} cycles")
        for blk in sorted(per):
            tot = sum(sum(v) / len(v) for v in per[blk].values())
            lines.append(f"  block {blk}: {tot:8.0f} cycles")
            for k in sorted(per[blk]):
                v = per[blk][k]
                lines.append(f"     {k:58s} mean {sum(v) / len(v):8.0f}  min {min(v):7d}  max {max(v):7d}")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

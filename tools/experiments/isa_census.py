#!/usr/bin/env python
"""Instruction census of a kernel from hipcc's -S output: per basic block, counts of MFMA / plain VALU /
transcendental / packed / accvgpr moves / LDS / SALU / VMEM instructions.  Used for the VALU-per-MFMA budget of
k_denoise_pipe.

    python tools/experiments/isa_census.py [kernel-name-substring] [--src difffacto_amd/csrc/denoiser_kernel.hip] [--blocks]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

TRANS = ("v_exp", "v_rcp", "v_log", "v_sin", "v_cos", "v_rsq", "v_sqrt")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accmov"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_mov") or op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_setprio"):
        return op.split("_")[1] if not op.startswith("s_waitcnt") else "waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", nargs="?", default="k_denoise_pipe")
    ap.add_argument("--src", default=os.path.join(ROOT, "difffacto_amd/csrc/denoiser_kernel.hip"))
    ap.add_argument("--asm", default=None, help="existing .s file")
    ap.add_argument("--blocks", action="store_true", help="print every basic block with >= 4 MFMAs or >= 40 instructions")
    ap.add_argument("--flags", default="")
    args = ap.parse_args()
    asm = args.asm
    if asm is None:
        from difffacto_amd import build as b
        asm = os.path.join(tempfile.gettempdir(), "isa_census.s")
        cmd = [b._hipcc(), *b.FLAGS, *args.flags.split(), "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm, args.src]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + args.kernel + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in lines[i])
    blocks = []
    cur = ("entry", collections.Counter(), start)
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), collections.Counter(), i)
            continue
        l = l.strip()
        if not l or l.startswith((";", ".")):
            continue
        op = l.split()[0]
        cur[1][classify(op)] += 1
        cur[1]["op:" + op] += 1
    blocks.append(cur)
    tot = collections.Counter()
    for _, c, _ in blocks:
        tot.update(c)
    keys = ["mfma", "valu", "pk", "cvt", "trans", "mov", "accmov", "lds", "salu", "waitcnt", "nop", "barrier", "vmem"]
    print("block".ljust(14) + "line".rjust(7) + "".join(k.rjust(8) for k in keys) + "   VALU/MFMA")
    for name, c, ln in blocks:
        n = sum(v for k, v in c.items() if not k.startswith("op:"))
        if args.blocks and (c["mfma"] >= 4 or n >= 40):
            v = c["valu"] + c["pk"] + c["cvt"] + c["trans"] + c["mov"] + c["accmov"]
            print(name.ljust(14) + str(ln - start).rjust(7) + "".join(str(c[k]).rjust(8) for k in keys) +
                  ("   %.2f" % (v / c["mfma"]) if c["mfma"] else ""))
    v = tot["valu"] + tot["pk"] + tot["cvt"] + tot["trans"] + tot["mov"] + tot["accmov"]
    print("TOTAL(static)".ljust(21) + "".join(str(tot[k]).rjust(8) for k in keys))
    if args.blocks:
        for name, c, ln in blocks:
            if c["mfma"] >= 16:
                ops = sorted(((k[3:], n) for k, n in c.items() if k.startswith("op:") and k[3:].startswith("v_") and "mfma" not in k), key=lambda t: -t[1])
                print(name, " ".join(f"{k}:{n}" for k, n in ops[:40]))


if __name__ == "__main__":
    main()

"""Builds libdfx.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m difffacto_amd.build [--force]

The .so lands in difffacto_amd/ (git-ignored, travels to the GPU box with the snapshot).
hipcc cross-compiles for gfx950 without a GPU.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libdfx.so")
BUILD = os.path.join(HERE, "_build")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", "-fno-slp-vectorize"]  # contraction is spelled explicitly (fmaf) where wanted: parity with the oracle


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return SO
    os.makedirs(BUILD, exist_ok=True)
    objs = []
    procs = []
    hdr_t = max(os.path.getmtime(p) for p in _deps() if not p.endswith(".hip"))
    for src in sources():
        obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [_hipcc(), *FLAGS, *extra_flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)

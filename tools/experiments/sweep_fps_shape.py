"""Workgroup shape of the register-resident FPS kernel (threads x points per thread) against the cloud size: time per call for
128 clouds, every shape that holds the cloud.  `python tools/experiments/sweep_fps_shape.py` -> profiles/r02_fps_shape_sweep.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from difffacto_amd import _ffi
from difffacto_amd.pointnet2_ops import pointnet2_utils as pu


def timeit(f, iters=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for N, M in ((16384, 4096), (8192, 2048), (4096, 1024), (2048, 512), (1024, 256), (512, 128)):
    xyz = torch.rand(128, N, 3, device="cuda")
    _ffi.lib().dfx_debug_fps_shape(0, 0)
    ref = pu.furthest_point_sample(xyz, M)
    row = [f"auto {timeit(lambda: pu.furthest_point_sample(xyz, M)):8.1f} us"]
    for nt in (256, 512, 1024):
        for ppt in (2, 4, 8, 16, 32):
            if nt * ppt < N or nt * ppt > 4 * N or (ppt == 32 and nt != 256):
                continue
            _ffi.lib().dfx_debug_fps_shape(nt, ppt)
            assert torch.equal(pu.furthest_point_sample(xyz, M), ref)
            row.append(f"{nt}x{ppt} {timeit(lambda: pu.furthest_point_sample(xyz, M)):8.1f}")
    _ffi.lib().dfx_debug_fps_shape(0, 0)
    print(f"N={N:5d} -> {M:4d}: " + " | ".join(row))

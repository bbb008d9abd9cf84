"""PointNetV2 training forward + backward alone (fp32 trunk), HIP-event time per iteration for each dfx_debug_bn_fused_stats mode:
python tools/experiments/time_pointnet_v2_train.py [B] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from difffacto_amd import _ffi, synth, training

MODES = (0, 1)   # (with tools/patches/r04_bn_bwd_sums_in_dx_epilogue.patch applied: (0, 2, 1) = separate passes / forward statistics only / + backward sums)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rng = np.random.Generator(np.random.PCG64(0))
W = synth.make_pointnet_v2_weights(0)
P = {n: torch.from_numpy(W[n].copy()).cuda().requires_grad_(True) for n in training.PNV2_PARAMS}
Bf = {n: torch.from_numpy(W[n].copy()).cuda() for n in training.PNV2_BUFFERS}
x = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)).cuda()
attn = torch.from_numpy(np.eye(4, dtype=np.float32)[rng.integers(0, 4, size=(B, N))]).cuda()
dm = torch.from_numpy(rng.standard_normal((B, 4, 256)).astype(np.float32)).cuda()
dv = torch.from_numpy(rng.standard_normal((B, 4, 256)).astype(np.float32)).cuda()


def it():
    for p in P.values():
        p.grad = None
    m, v = training.pointnet_v2_train_forward(P, Bf, x, attn, momentum=0.1, precision="f32")
    ((m * dm).sum() + (v * dv).sum()).backward()


for rnd in range(3):
    for mode in MODES:
        _ffi.lib().dfx_debug_bn_fused_stats(mode)
        for _ in range(3):
            it()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            it()
        b.record()
        torch.cuda.synchronize()
        print(f"round {rnd} bn_fused_stats({mode}): {a.elapsed_time(b) / 20:.3f} ms per forward + backward (B={B} N={N})")
_ffi.lib().dfx_debug_bn_fused_stats(1)

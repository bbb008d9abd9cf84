"""Prior loss through the 4 x 14 coupling layers, forward + backward, wall time per call vs batch size (GPU box): python tools/experiments/time_prior_loss.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from difffacto_amd import synth, training

W = synth.make_latent_weights(seed=0)
P = {k: torch.from_numpy(v.copy()).cuda().requires_grad_(True) for k, v in W.items() if k.startswith("flow.")}
for B in (8, 16, 32, 64, 128):
    rng = np.random.Generator(np.random.PCG64(B))
    z = torch.from_numpy(rng.standard_normal((B, 256, 4)).astype(np.float32)).cuda().requires_grad_(True)
    lv = torch.from_numpy((0.1 * rng.standard_normal((B, 4, 256))).astype(np.float32)).cuda().requires_grad_(True)
    valid = torch.ones(B, 4, device="cuda")

    def it():
        for p in P.values():
            p.grad = None
        loss, _, _ = training.prior_loss(P, z, lv, valid)
        loss.backward()

    for _ in range(3):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        it()
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per forward + backward")

"""Denoiser training forward + loss + backward at B=128 x 2048 (bf16 products): eager launches vs one captured hipGraph replayed
(python tools/experiments/time_train_graph.py).  The step is GPU-bound with no idle gaps; the question is what the ~50 small launches cost inside a graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from difffacto_amd import synth, training

B, N = 128, 2048
rng = np.random.Generator(np.random.PCG64(0))
W = synth.make_denoiser_weights(0)
pc, mean, logvar, valid = synth.make_latents(B, seed=1)
seg = synth.make_seg_mask(valid, N)
var = np.exp(logvar).astype(np.float32)
idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
anc, vr = np.take_along_axis(mean, idx, axis=2), np.take_along_axis(var, idx, axis=2)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = {k: cu(v).requires_grad_(True) for k, v in W.items()}
args = [cu((anc + np.sqrt(vr) * rng.standard_normal((B, 3, N))).astype(np.float32)), cu(rng.integers(0, 1000, size=(B,)).astype(np.int32)), cu(pc),
        cu(np.concatenate([mean, var], 1).astype(np.float32)), cu(anc.transpose(0, 2, 1)), cu(vr.transpose(0, 2, 1)), cu(valid), cu(seg.astype(np.int32))]
noise = cu(rng.standard_normal((B, 3, N)).astype(np.float32))


def step():
    for p in P.values():
        p.grad = None
    loss = training.masked_mse(noise, training.denoiser_train_forward(P, *args, precision="bf16"), None)
    loss.backward()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for rnd in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            step()
        b.record()
        torch.cuda.synchronize()
        print(f"round {rnd}: eager  {a.elapsed_time(b) / 20:.3f} ms per forward + loss + backward")
        if rnd == 0:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
        a.record()
        for _ in range(20):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        print(f"round {rnd}: graph  {a.elapsed_time(b) / 20:.3f} ms per replay")

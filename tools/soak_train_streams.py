"""Race soak of the training step's side stream: the same iteration repeated with dfx_debug_train_streams(1) must give the single-stream bits
every time (a missing dependency between the two streams would show up as a run-to-run difference): python tools/soak_train_streams.py [reps] [B] [N] [dropout p]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import _ffi, synth, training

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
DROP = (float(sys.argv[4]), 4242) if len(sys.argv) > 4 else None   # e.g. 0.2: the shipped configuration (same Philox key every repetition)
rng = np.random.Generator(np.random.PCG64(7))
W = synth.make_denoiser_weights(0)
pc, mean, logvar, valid = synth.make_latents(B, seed=3, all_valid=False)
seg = synth.make_seg_mask(valid, N)
var = np.exp(logvar).astype(np.float32)
idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
anc, vr = np.take_along_axis(mean, idx, axis=2), np.take_along_axis(var, idx, axis=2)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = {k: cu(v).requires_grad_(True) for k, v in W.items()}
cc, cm = cu(pc).requires_grad_(True), cu(np.concatenate([mean, var], 1).astype(np.float32)).requires_grad_(True)
args = [cu((anc + np.sqrt(vr) * rng.standard_normal((B, 3, N))).astype(np.float32)), cu(rng.integers(0, 1000, size=(B,)).astype(np.int32)), cc, cm,
        cu(anc.transpose(0, 2, 1)), cu(vr.transpose(0, 2, 1)), cu(valid), cu(seg.astype(np.int32))]
noise = cu(rng.standard_normal((B, 3, N)).astype(np.float32))


def step():
    for p in list(P.values()) + [cc, cm]:
        p.grad = None
    eps = training.denoiser_train_forward(P, *args, precision="bf16", dropout=DROP)
    training.masked_mse(noise, eps, None).backward()
    torch.cuda.synchronize()
    return [eps.detach().clone()] + [p.grad.clone() for p in P.values()] + [cc.grad.clone(), cm.grad.clone()]


_ffi.lib().dfx_debug_train_streams(0)
ref = step()
_ffi.lib().dfx_debug_train_streams(1)
bad = 0
for r in range(reps):
    got = step()
    if not all(torch.equal(a, b) for a, b in zip(got, ref)):
        bad += 1
print(f"side-stream soak B={B} N={N} dropout={DROP}: {reps} iterations against the single-stream bits, {bad} different")
sys.exit(1 if bad else 0)

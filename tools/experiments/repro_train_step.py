"""Run-to-run reproducibility of one training iteration: python tools/experiments/repro_train_step.py B N

Four runs of the same iteration per (dropout, stream mode); prints how many of eps / parameter gradients / context gradients differ in ANY bit between
runs, then a per-tensor SAME / DIFF table for two Dropout(0.2) runs.  This is the script that located round 5's stale dropout-bit-word race in
k_ff<true, true> (HISTORY.md round 6, item 11): eps and the last block's gradients were the same, everything below it differed.  Expected now: 0 everywhere.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from difffacto_amd import _ffi, synth, training
B, N = int(sys.argv[1]), int(sys.argv[2])
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
names = ["eps"] + list(P) + ["cc", "cm"]
def step(drop):
    for p in list(P.values()) + [cc, cm]: p.grad = None
    eps = training.denoiser_train_forward(P, *args, precision="bf16", dropout=drop)
    training.masked_mse(noise, eps, None).backward()
    torch.cuda.synchronize()
    return [eps.detach().clone()] + [p.grad.clone() for p in P.values()] + [cc.grad.clone(), cm.grad.clone()]
def diff(a, b):
    return [(n, float((x - y).abs().max()), float(y.abs().max())) for n, x, y in zip(names, a, b) if not torch.equal(x, y)]
for drop in (None, (0.2, 4242)):
    for streams in (0, 1):
        _ffi.lib().dfx_debug_train_streams(streams)
        r = [step(drop) for _ in range(4)]
        d01, d12 = diff(r[0], r[1]), diff(r[1], r[2])
        print(f"dropout {drop} streams {streams}: run0 vs run1 differ in {len(d01)} tensors, run1 vs run2 in {len(d12)}", d01[:4])
        if streams == 0: ref = r[1]
        else: print("   streams 1 vs streams 0:", diff(r[1], ref)[:6])
_ffi.lib().dfx_debug_train_streams(0)
a, b = step((0.2, 4242)), step((0.2, 4242))
for n, x, y in zip(names, a, b):
    print(f"{'SAME ' if torch.equal(x, y) else 'DIFF '} {n:50s} {float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)):.2e}")

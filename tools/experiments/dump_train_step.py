"""One training iteration's eps and gradients into an .npz, for bit comparisons across builds: python tools/experiments/dump_train_step.py out.npz B N [dropout p]
   and the comparison itself:                                                                python tools/experiments/dump_train_step.py --cmp a.npz b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [(k, float(np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30))) for k in a.files if not np.array_equal(a[k], b[k])]
    print(f"{sys.argv[2]} vs {sys.argv[3]}: {len(a.files)} tensors, {len(bad)} differ", bad[:6])
    sys.exit(1 if bad else 0)
import torch
from difffacto_amd import synth, training
out, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
drop = (float(sys.argv[4]), 4242) if len(sys.argv) > 4 else None
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
eps = training.denoiser_train_forward(P, *args, precision="bf16", dropout=drop)
training.masked_mse(noise, eps, None).backward()
torch.cuda.synchronize()
d = {"eps": eps.detach().cpu().numpy(), "cc": cc.grad.cpu().numpy(), "cm": cm.grad.cpu().numpy()}
d.update({"g/" + k: v.grad.cpu().numpy() for k, v in P.items()})
np.savez(out, **d)
print(f"{out}: B={B} N={N} dropout={drop}, {len(d)} tensors, finite {all(np.isfinite(v).all() for v in d.values())}")

#!/usr/bin/env python
"""Golden vectors for the PointNet++ ENCODER classes (python/difffacto/models/encoders/pointnet2.py): PointNet2SSG / PointNet2MSG.

    python tests/golden/make_golden_pn2enc.py          # dev container only; writes tests/golden/pn2enc_*.npz

Runs the REFERENCE's own classes on CPU: `pointnet2_ops` = the reference's Python package over the C-oracle shim of make_golden_sa.py (nvcc is absent), the
encoder module re-executed on top of it.  eval() forward for both classes; for PointNet2SSG also one train() step (batch-statistics BatchNorm, autograd through
grouping / conv / max / the Linear head; B = 16 so that the head's BatchNorm1d over the batch is well conditioned: with B = 4 torch-on-GPU and
torch-on-CPU already disagree by 5e-3 on these gradients) with `fc_layer[6].p = 0` — nn.Dropout(0.5) draws from torch's generator, which no other implementation can replay.
Large gradients are stored as 512 sampled entries + (sum, L2 norm), like train_grads_*.npz (tests/_train_case.check_against_golden).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
F32 = np.float32


def reference_encoders():
    import ref_import
    ref_import.import_reference()                       # difffacto importable (stubs for the CUDA extensions)
    import make_golden_sa
    make_golden_sa.install_ext_shim()                   # pointnet2_ops := the reference's Python package over the C oracle
    from difffacto.utils.registry import ENCODERS
    for k in ("PointNet2SSG", "PointNet2MSG"):          # the module below registers them again
        ENCODERS._modules.pop(k, None)
    import difffacto.models.encoders.pointnet2 as m
    m = importlib.reload(m)                             # re-executed: binds the real PointnetSAModule(MSG) classes
    assert m.PointnetSAModule.__module__ == "pointnet2_ops.pointnet2_modules"
    return m, make_golden_sa.randomize


def pack_grads(out, grads, rng):
    for k, g in grads.items():
        g = g.astype(F32)
        if g.size <= 20000:
            out["g/" + k] = g.ravel()
        else:
            idx = np.sort(rng.choice(g.size, size=512, replace=False))
            out["gs/" + k], out["gi/" + k] = g.ravel()[idx], idx
            out["gn/" + k] = np.array([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum())])


def main():
    torch.manual_seed(0)
    m, randomize = reference_encoders()
    for cls, tag, B, N, seed in ((m.PointNet2SSG, "ssg", 16, 640, 61), (m.PointNet2MSG, "msg", 2, 640, 62)):
        rng = np.random.Generator(np.random.PCG64(seed))
        pc = np.concatenate([rng.uniform(-0.5, 0.5, size=(B, N, 3)), rng.standard_normal((B, N, 4)) * 0.5], axis=2).astype(F32)   # xyz + 4 extra channels
        enc = cls(additioinal_dim=4, zdim=16, num_anchors=4)
        W = randomize(enc, rng)
        enc.eval()
        with torch.no_grad():
            z = enc(torch.from_numpy(pc))
        # the 1.4 M weights are NOT stored: `randomize` draws them from this generator in state_dict order, the test repeats the draws on the mirror (same
        # module tree, same order) and checks the per-tensor sums kept here
        out = dict(pc=pc, z=z.numpy().astype(F32), zdim=np.array(16), seed=np.array(seed), wkeys=np.array(list(W)),
                   wsum=np.array([float(v.astype(np.float64).sum()) for v in W.values()]))
        if tag == "ssg":   # one training step (Dropout switched off: see the module docstring)
            enc.train()
            enc.fc_layer[6].p = 0.0
            t = torch.from_numpy(pc).requires_grad_(True)
            zt = enc(t)
            gout = rng.standard_normal(tuple(zt.shape)).astype(F32)
            zt.backward(torch.from_numpy(gout))
            out["z_train"], out["gout"], out["d_pc"] = zt.detach().numpy().astype(F32), gout, t.grad.numpy().astype(F32)
            pack_grads(out, {k: p.grad.numpy() for k, p in enc.named_parameters()}, rng)
            out.update({"after." + k: v.numpy().copy() for k, v in enc.state_dict().items() if "running_" in k})
        np.savez_compressed(os.path.join(HERE, f"pn2enc_{tag}.npz"), **out)
        print("wrote pn2enc_" + tag, z.shape, float(z.abs().max()), {k: v.shape for k, v in list(out.items())[:3]})


if __name__ == "__main__":
    main()

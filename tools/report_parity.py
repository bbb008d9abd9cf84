"""bf16-vs-fp32 parity of the persistent chain on identical explicit noise (GPU box):
max-abs / mean-abs point deviation and the Chamfer-L2 distance between the two generated clouds, next to the
Chamfer-L2 between two independent samples of the same shape (the natural scale).  Writes one line per T."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import synth
from difffacto_amd.engine import DenoiserEngine
from difffacto_amd.metrics import EMD, chamfer_l2

B, N = 4, 2048
W = {k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()}
pc, m, lv, va = synth.make_latents(B, seed=5)
args = tuple(map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
seg = torch.from_numpy(synth.make_seg_mask(va, N))
for T in (100, 1000):
    g = torch.Generator().manual_seed(T)
    xT = torch.randn(B, 3, N, generator=g)
    zs = torch.randn(T, B, 3, N, generator=g)
    zs2 = torch.randn(T, B, 3, N, generator=g)
    out = {}
    for prec in ("f32", "bf16"):
        eng = DenoiserEngine(W, T, precision=prec)
        ctx = eng.prepare_shapes(*args)
        out[prec], _ = eng.sample_chain(ctx, seg, x_T_noise=xT, step_noise=zs)
        if prec == "f32":
            other, _ = eng.sample_chain(ctx, seg, x_T_noise=xT.flip(0), step_noise=zs2)
        eng.close()
    d = (out["f32"] - out["bf16"]).abs()
    cd = chamfer_l2(out["f32"], out["bf16"])
    cd_ind = chamfer_l2(out["f32"], other)

    def emd_pair(x, y):   # the evaluation's normalisation-free call (datasets/evaluation_utils.py:84-89) on clouds scaled into [0,1]^3
        lo = torch.minimum(x.amin((1, 2), keepdim=True), y.amin((1, 2), keepdim=True))
        hi = torch.maximum(x.amax((1, 2), keepdim=True), y.amax((1, 2), keepdim=True))
        return EMD(0.002, 10000, True)((x - lo) / (hi - lo), (y - lo) / (hi - lo))
    emd, emd_ind = emd_pair(out["f32"], out["bf16"]), emd_pair(out["f32"], other)
    print(f"T={T} B={B} N={N}: max-abs {d.max().item():.3e}  mean-abs {d.mean().item():.3e}  "
          f"Chamfer-L2(bf16, fp32) mean {cd.mean().item():.3e} max {cd.max().item():.3e}  |  "
          f"Chamfer-L2 between two independent fp32 samples {cd_ind.mean().item():.3e}  "
          f"(part sigma ~{float(np.sqrt(np.exp(lv)).mean()):.3f})  |  EMD(bf16, fp32) on the unit-box-normalised clouds "
          f"{emd.mean().item():.3e} vs {emd_ind.mean().item():.3e} between independent samples")

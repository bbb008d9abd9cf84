import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine
W = {k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()}
for N in (128, 256):
    B = 2
    pc, m, lv, va = synth.make_latents(B, seed=1)
    seg = torch.from_numpy(synth.make_seg_mask(va, N))
    x = torch.randn(B, 3, N, generator=torch.Generator().manual_seed(0))
    out = {}
    for prec in ("f32", "bf16"):
        e = DenoiserEngine(W, 10, precision=prec)
        ctx = e.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
        out[prec] = e.eps(ctx, x, seg, 5).cpu()
    d = (out["f32"] - out["bf16"]).abs()
    print(N, "max err", float(d.max()), "nan", bool(torch.isnan(out["bf16"]).any()), "ref max", float(out["f32"].abs().max()))

"""Soak test of the pipelined chain's LDS ring protocol (GPU box): the same launch repeated must be bit-identical.
A protocol race (a ring slot overwritten before its last reader, a read ahead of its DMA) would show up as a mismatch.
    python tools/soak_determinism.py [reps] [T] [bf16|f32] [pipe-waves]      (pipe-waves 64 = k_denoise_pipe2)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import synth
from difffacto_amd.engine import DenoiserEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
if len(sys.argv) > 4:
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_pipe_waves(int(sys.argv[4]))
W = {k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()}
bad = 0
for B, N in ((128, 2048), (32, 8192), (37, 2048)):
    pc, m, lv, va = synth.make_latents(B, seed=B)
    eng = DenoiserEngine(W, T, precision=prec)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32) * 0.05, va)))
    seg = torch.from_numpy(synth.make_seg_mask(va, N))
    ref, _ = eng.sample_chain(ctx, seg, seed=11)
    assert torch.isfinite(ref).all()
    for r in range(reps):
        out, _ = eng.sample_chain(ctx, seg, seed=11)
        if not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH B={B} N={N} rep {r}: max abs {float((out - ref).abs().max()):.3e}, {int((out != ref).sum())} values")
    print(f"B={B} N={N} T={T}: {reps} repetitions compared, |x|max {float(ref.abs().max()):.3f}")
    eng.close()
print(f"soak [{prec}{' pipe-waves ' + sys.argv[4] if len(sys.argv) > 4 else ''}]:", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)

"""bf16 chain (k_denoise_pipe<8>, forced) vs the exact-fp32 HIP chain on identical explicit noise: the yardstick for kernel experiments that
change the bf16 arithmetic (tools/experiments/r04_headline_experiments.sh).  python tools/experiments/exp_parity.py [T] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from difffacto_amd import _ffi, synth
from difffacto_amd.engine import DenoiserEngine, last_kernel_variant

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = 2048
W = {k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(seed=0).items()}
pc, mean, logvar, valid = synth.make_latents(B, seed=5)
var = np.exp(logvar).astype(np.float32)
seg = torch.from_numpy(synth.make_seg_mask(valid, N))
g = torch.Generator(device="cuda").manual_seed(T)
xT = torch.randn(B, 3, N, device="cuda", generator=g)
zs = torch.randn(T, B, 3, N, device="cuda", generator=g)
x1 = torch.randn(B, 3, N, device="cuda", generator=g)
out, eps = {}, {}
for prec in ("f32", "bf16"):
    eng = DenoiserEngine(W, num_timesteps=T, precision=prec)
    ctx = eng.prepare_shapes(*(torch.from_numpy(a) for a in (pc, mean, var, valid)))
    _ffi.lib().dfx_debug_pipe_waves(8)
    out[prec], _ = eng.sample_chain(ctx, seg, x_T_noise=xT, step_noise=zs)
    name = last_kernel_variant()
    eps[prec] = eng.eps(ctx, x1, seg, T // 2)
    _ffi.lib().dfx_debug_pipe_waves(0)
    eng.close()
d = (out["bf16"] - out["f32"]).abs()
e = (eps["bf16"] - eps["f32"]).abs()
print(f"T={T} B={B} {name}: chain bf16 vs f32 max-abs {d.max().item():.3e} mean-abs {d.mean().item():.3e}; one evaluation: max-abs {e.max().item():.3e} "
      f"rms {e.pow(2).mean().sqrt().item():.3e} (|eps| max {eps['f32'].abs().max().item():.2f})")

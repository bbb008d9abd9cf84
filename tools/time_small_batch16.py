"""B = 1 .. 32 shapes of 2048 points, T = 1000: HIP-event time of one chain launch for every chain-kernel variant (forced) and the launcher's own choice.
python tools/time_small_batch16.py  ->  profiles/r05_small_batch_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import _ffi, synth
from difffacto_amd.engine import DenoiserEngine, last_kernel_variant

T, N = 1000, 2048
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()}, T, precision="bf16")
VAR = {"coop16": 160, "coop": 1, "coop2": 16, "pipe<2>": 2, "pipe<4>": 4, "pipe<8>": 8, "auto": 0}
print(f"{'B':>3} " + " ".join(f"{k:>10}" for k in VAR) + "   (ms per T = 1000 chain; auto: the launcher's choice)")
for B in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
    pc, mean, logvar, valid = synth.make_latents(B, seed=1)
    ctx = eng.prepare_shapes(*(torch.from_numpy(a) for a in (pc, mean, np.exp(logvar).astype(np.float32), valid)))
    seg = torch.from_numpy(synth.make_seg_mask(valid, N))
    row, chosen = [], ""
    for name, code in VAR.items():
        if (name == "coop2" and B > 16) or (name in ("coop", "coop16") and B > 16):
            row.append(float("nan"))
            continue
        _ffi.lib().dfx_debug_pipe_waves(code)
        eng.eps(ctx, torch.zeros(B, 3, N), seg, 5)          # warm-up of the same kernel
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pred, _ = eng.sample_chain(ctx, seg, seed=B)
        b.record()
        torch.cuda.synchronize()
        row.append(a.elapsed_time(b))
        if name == "auto":
            chosen = last_kernel_variant()
        assert torch.isfinite(pred).all()
    _ffi.lib().dfx_debug_pipe_waves(0)
    print(f"{B:>3} " + " ".join(f"{v:>10.2f}" for v in row) + f"   auto = {chosen}; {B / min(v for v in row if v == v) * 1e3:.1f} shapes/s at the fastest")

"""HIP-event time of dfx_shape_ctx_prepare (k_shape_ctx) alone: python tools/experiments/time_shape_ctx.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from difffacto_amd import synth
from difffacto_amd.engine import DenoiserEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
W = synth.make_denoiser_weights(seed=0)
eng = DenoiserEngine({k: torch.from_numpy(v).to(dev) for k, v in W.items()}, num_timesteps=100, precision="bf16", device=dev)
pc, mean, logvar, valid = synth.make_latents(B, seed=1)
pc, mean, var, valid = (torch.from_numpy(x).to(dev) for x in (pc, mean, (logvar * 0 + 1).astype("float32"), valid))
for _ in range(5):
    eng.prepare_shapes(pc, mean, var, valid)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(100):
    eng.prepare_shapes(pc, mean, var, valid)
b.record()
torch.cuda.synchronize()
print(f"dfx_shape_ctx_prepare B={B}: {a.elapsed_time(b) * 10:.1f} us per call")

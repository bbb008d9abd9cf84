"""The generation front end (draws + dfx_sample_latents + dfx_shape_ctx_prepare) launch by launch: python tools/experiments/trace_front_end.py [B] [N] [T]
Run under rocprofv3 --kernel-trace and feed the CSV to tools/experiments/iter_sequence.py with the chain kernel as the marker (k_denoise)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from difffacto_amd import synth
from difffacto_amd.engine import DenoiserEngine
from difffacto_amd.latents import LatentSampler

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
W = synth.make_denoiser_weights(seed=0)
L = synth.make_latent_weights(seed=0)
eng = DenoiserEngine({k: torch.from_numpy(v).to(dev) for k, v in W.items()}, num_timesteps=T, precision="bf16", device=dev)
sampler = LatentSampler({k: torch.from_numpy(v).to(dev) for k, v in L.items()}, noise_scale=100.0, device=dev)
valid = torch.from_numpy(synth.make_latents(B, seed=1000)[3]).to(dev)
for i in range(4):
    w = torch.randn(B, 256, 4, device=dev)
    an = torch.randn(B, 32, device=dev)
    lat = sampler.sample_latents(w, an, valid, K=1, npoints=N)
    ctx = eng.prepare_shapes(lat["part_code"], lat["params"][:, :3], lat["params"][:, 3:], lat["valid_id"])
    pred, _ = eng.sample_chain(ctx, lat["seg_mask"], seed=7000 + i)
torch.cuda.synchronize()
print("ok", tuple(pred.shape))

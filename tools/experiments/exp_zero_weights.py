"""Timing diagnostic: the T=50 chain at B=128 with (a) the synthetic weights, (b) all denoiser weights zero, (c) weights quantised
to few distinct values — same instruction stream, different operand toggling (power)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from difffacto_amd import synth
from difffacto_amd.engine import DenoiserEngine
T, B, N = 50, 128, 2048
W0 = synth.make_denoiser_weights(0)
pc, m, lv, va = synth.make_latents(B, seed=1)
seg = torch.from_numpy(synth.make_seg_mask(va, N))
def run(tag, W, pcz=False):
    eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision="bf16")
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc * (0 if pcz else 1), m, np.exp(lv).astype(np.float32), va)))
    eng.sample_chain(ctx, seg, seed=1)
    ev = []
    for i in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.sample_chain(ctx, seg, seed=2 + i); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    print(f"{tag}: {np.mean([a.elapsed_time(b) for a, b in ev]):.3f} ms")
    eng.close()
run("synthetic weights", W0)
Wz = {k: np.zeros_like(v) for k, v in W0.items()}
for k in Wz:
    if "norm" in k and k.endswith("weight"): Wz[k] = np.ones_like(Wz[k])
run("all-zero weights", Wz, pcz=True)
Wf = dict(W0)
for k in Wf:
    if ".ff.net" in k and k.endswith("weight"): Wf[k] = np.zeros_like(Wf[k])
run("zero FF weights only (W1, W2)", Wf)
Wq = dict(W0)
for k in Wq:
    if ".ff.net" in k and k.endswith("weight"): Wq[k] = (np.sign(Wq[k]) * 0.0625).astype(np.float32)
run("FF weights = +-1/16 (one mantissa pattern)", Wq)
run("synthetic weights again", W0)

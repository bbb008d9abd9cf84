"""Run-to-run reproducibility of the headline launch: the T = 1000 chain at B = 128 x 2048 (k_denoise_pipe<8>, in-kernel Philox noise keyed by an explicit seed)
repeated and compared bit by bit with the first run: python tools/soak_headline_chain.py [reps] [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from difffacto_amd import engine, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
N = 2048
W = synth.make_denoiser_weights(seed=0)
pc, mean, logvar, valid = synth.make_latents(B, seed=5)
var = np.exp(logvar).astype(np.float32)
seg = synth.make_seg_mask(valid, N)
bad = {}
for prec in ("bf16", "f32") if T <= 100 else ("bf16",):
    eng = engine.DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision=prec)
    sc = eng.prepare_shapes(*map(torch.from_numpy, (pc, mean, var, valid)))
    run = lambda: eng.sample_chain(sc, torch.from_numpy(seg), seed=20260930)[0].clone()
    ref = run()
    bad[prec] = sum(not torch.equal(run(), ref) for _ in range(reps))
    print(f"{prec}: kernel {engine.last_kernel_variant()}, finite {bool(torch.isfinite(ref).all())}, {reps} repetitions, {bad[prec]} different")
    eng.close()
sys.exit(1 if any(bad.values()) else 0)

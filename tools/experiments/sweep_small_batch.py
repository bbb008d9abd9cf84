"""Small-batch latency of the T = 1000 chain (N = 2048, bf16): the co-operative kernels (one and two tiles per workgroup) against the pipelined one
at 2 / 4 / 8 wavefronts per workgroup.  python tools/experiments/sweep_small_batch.py [B ...]   (prints ms per chain and shapes/s; picks the crossover for launch())"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine

T, N = 1000, 2048
W = synth.make_denoiser_weights(0)
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision="bf16")
batches = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 6, 8, 10, 12, 16, 24, 32]
print("%4s " % "B" + "".join("%22s" % n for n in ("auto", "coop", "coop2", "pipe<2>", "pipe<4>", "pipe<8>")))
for B in batches:
    pc, m, lv, va = synth.make_latents(B, seed=1)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
    seg = torch.from_numpy(synth.make_seg_mask(va, N))
    row = []
    for nw in (0, 1, 16, 2, 4, 8):
        _ffi.lib().dfx_debug_pipe_waves(nw)
        eng.sample_chain(ctx, seg, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng.sample_chain(ctx, seg, seed=1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 500
        row.append("%9.2f ms %7.1f /s" % (ms, B / ms * 1e3))
    _ffi.lib().dfx_debug_pipe_waves(0)
    print("%4d " % B + " ".join("%21s" % r for r in row))

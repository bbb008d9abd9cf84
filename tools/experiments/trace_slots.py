"""Dump per-slot durations of the pipelined kernel (debug trace): python tools/experiments/trace_slots.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine
import os
T, B, N, CAP = 4, int(os.environ.get("DFX_TRACE_B", "128")), 2048, 2048
FLAGS = 0   # (the run-time ablation flags are gone: ablations are builds, tools/patches/)
W = synth.make_denoiser_weights(0)
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision="bf16")
pc, m, lv, va = synth.make_latents(B, seed=1)
ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
seg = torch.from_numpy(synth.make_seg_mask(va, N))
eng.sample_chain(ctx, seg, seed=1)
buf = torch.zeros(2 * CAP, dtype=torch.int64, device="cuda")
_ffi.lib().dfx_debug_trace(ctypes.c_void_p(buf.data_ptr()), CAP)
eng.sample_chain(ctx, seg, seed=1)
torch.cuda.synchronize()
_ffi.lib().dfx_debug_trace(None, 0)
tr = buf.cpu().numpy().reshape(2, CAP)
for g in range(2):
    t = tr[g]; t = t[t != 0]
    tag = ((t >> 56) & 0xff).astype(int); clk = (t & ((1 << 56) - 1)).astype(np.int64)
    ev = list(zip(tag.tolist(), (clk - clk[0]).tolist()))
    # print a window of raw events in FF steady state
    k0 = 260
    seq = ev[k0:k0 + 28]
    print(f"flags {FLAGS} group {'AB'[g]}: events (tag: 1=arrive at mgmt barrier, 2=released, 3=slot switch), deltas:")
    print("   ", [(a[0], a[1] - b[1]) for a, b in zip(seq[1:], seq[:-1])])
    print(f"    total {clk[-1] - clk[0]} ticks for {len(ev)} events")

"""Phase durations of the co-operative latency kernel (build with -DDFX_TRACE): python tools/trace_coop.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine
T, B, N, CAP = 6, 1, 2048, 4096
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
names = {10: "arrive b0", 11: "pass b0", 12: "arrive b1", 13: "pass b1", 14: "arrive b2", 15: "pass b2"}
for w in range(2):
    t = tr[w]; t = t[t != 0]
    tag = ((t >> 56) & 0xff).astype(int); clk = (t & ((1 << 56) - 1)).astype(np.int64)
    ev = list(zip(tag.tolist(), clk.tolist()))
    seq = ev[36:36 + 13]   # a steady-state window (blocks 6, 7 = second step)
    print("owner" if w == 0 else "helper wave 1", [(names[a[0]], a[1] - b[1]) for a, b in zip(seq[1:], seq[:-1])])
    print("   total", clk[-1] - clk[0], "ticks for", len(ev), "events =", (clk[-1] - clk[0]) / (len(ev) / 6), "per block")

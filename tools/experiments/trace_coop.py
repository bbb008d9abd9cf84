"""Phase durations of the co-operative latency kernel (build with -DDFX_TRACE): python tools/experiments/trace_coop.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
names = {10: "arrive b0", 11: "pass b0", 12: "arrive b1", 13: "pass b1", 14: "arrive b2", 15: "pass b2", 16: "arrive end", 17: "pass end", 18: "post_eps", 19: "epilogue", 20: "proj_in"}
for w in range(2):
    t = tr[w]; t = t[t != 0]
    tag = ((t >> 56) & 0xff).astype(int); clk = (t & ((1 << 56) - 1)).astype(np.int64)
    ev = list(zip(tag.tolist(), clk.tolist()))
    k0 = [i for i, e in enumerate(ev) if e[0] == 17][2] + (3 if w == 0 else 0)
    seq = ev[k0 - 16:k0 + 14]   # last block of a step, the step boundary, first block of the next
    print("owner" if w == 0 else "wave 5", [(names[a[0]], a[1] - b[1]) for a, b in zip(seq[1:], seq[:-1])])
    print("   total", clk[-1] - clk[0], "ticks for", len(ev), "events =", (clk[-1] - clk[0]) / (sum(1 for e in ev if e[0] == 10)), "per block")
    b0 = [c for g, c in ev if g == 10]
    print("   block periods (arrive b0 -> arrive b0):", np.diff(b0).tolist())

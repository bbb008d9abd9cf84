"""Phase trace of k_denoise_pipe_f32 (waves 0 and 4 of workgroup 0 = the two wavefronts of one SIMD): needs -DDFX_TRACE.
Tags: 1 = arrives at a record barrier, 2 = released, 3 = its DMA pieces of the record three ahead are issued."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine
T, B, N, CAP = 2, int(os.environ.get("DFX_TRACE_B", "128")), 2048, 4096
W = synth.make_denoiser_weights(0)
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision="f32")
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
base = None
for g in range(2):
    t = tr[g]; t = t[t != 0]
    tag = ((t >> 56) & 0xff).astype(int); clk = (t & ((1 << 56) - 1)).astype(np.int64)
    if base is None: base = clk[0]
    ev = list(zip(tag.tolist(), (clk - base).tolist()))
    k0 = 3 * 60
    seq = ev[k0:k0 + 25]
    print(f"wave {4 * g}: {len(ev)} events, span {ev[-1][1] - ev[0][1]} cycles; window (tag, absolute cycle, delta):")
    print("   ", [(a[0], a[1], a[1] - b[1]) for a, b in zip(seq[1:], seq[:-1])])
    # totals per tag transition over a whole block (36 records = 108 events)
    d = {}
    for a, b in zip(ev[109:109 + 108], ev[108:108 + 108]):
        d[(b[0], a[0])] = d.get((b[0], a[0]), 0) + a[1] - b[1]
    print("    one block:", {f"{k[0]}->{k[1]}": v for k, v in d.items()}, "sum", sum(d.values()))

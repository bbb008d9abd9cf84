"""Phase trace of k_denoise_pipe2 (wave 0 of workgroup 0): needs a library built with -DDFX_TRACE.
   python tools/experiments/trace_pipe2.py   ->  cycles per phase of one transformer block in steady state"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine
T, B, N, CAP = 3, int(os.environ.get("DFX_TRACE_B", "128")), 2048, 4096
W = synth.make_denoiser_weights(0)
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision="bf16")
pc, m, lv, va = synth.make_latents(B, seed=1)
ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
seg = torch.from_numpy(synth.make_seg_mask(va, N))
_ffi.lib().dfx_debug_pipe_waves(64)
eng.sample_chain(ctx, seg, seed=1)
buf = torch.zeros(2 * CAP, dtype=torch.int64, device="cuda")
_ffi.lib().dfx_debug_trace(ctypes.c_void_p(buf.data_ptr()), CAP)
eng.sample_chain(ctx, seg, seed=1)
torch.cuda.synchronize()
_ffi.lib().dfx_debug_trace(None, 0)
t = buf.cpu().numpy()[:CAP]
t = t[t != 0]
tag = ((t >> 56) & 0xff).astype(int)
clk = (t & ((1 << 56) - 1)).astype(np.int64)
ev = list(zip(tag.tolist(), (clk - clk[0]).tolist()))
print(f"{len(ev)} events, total {ev[-1][1]} ticks")
# one block = the events between two attention-record barriers: find tags 21 (after M0) and cut there
cuts = [i for i, e in enumerate(ev) if e[0] == 21]
for a, b in list(zip(cuts, cuts[1:]))[6:9]:
    blk = ev[a:b + 1]
    print("block:", blk[-1][1] - blk[0][1], "ticks;", [(x[0], x[1] - y[1]) for x, y in zip(blk[1:], blk[:-1])])

"""Phase durations of the co-operative kernels (library built with -DDFX_TRACE): python tools/trace_coop16.py [160|1]   (160 = 16-point tiles, 1 = 32-point)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"])
from difffacto_amd import synth, _ffi
from difffacto_amd.engine import DenoiserEngine, last_kernel_variant
code = int(sys.argv[1]) if len(sys.argv) > 1 else 160
T, B, N, CAP = 6, 1, 2048, 4096
eng = DenoiserEngine({k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()}, num_timesteps=T, precision="bf16")
pc, m, lv, va = synth.make_latents(B, seed=1)
ctx = eng.prepare_shapes(*map(torch.from_numpy, (pc, m, np.exp(lv).astype(np.float32), va)))
seg = torch.from_numpy(synth.make_seg_mask(va, N))
_ffi.lib().dfx_debug_pipe_waves(code)
eng.sample_chain(ctx, seg, seed=1)
buf = torch.zeros(2 * CAP, dtype=torch.int64, device="cuda")
_ffi.lib().dfx_debug_trace(ctypes.c_void_p(buf.data_ptr()), CAP)
eng.sample_chain(ctx, seg, seed=1)
torch.cuda.synchronize()
print("kernel:", last_kernel_variant())
_ffi.lib().dfx_debug_trace(None, 0)
tr = buf.cpu().numpy().reshape(2, CAP)
names = {10: "arrive b0", 11: "pass b0 [top: waits + W1 gathers]", 12: "arrive b1 [phase A]", 13: "pass b1", 14: "arrive b2 [phase H]", 15: "pass b2", 16: "arrive end [phase G]", 17: "pass end",
         18: "post_eps", 19: "epilogue", 20: "proj_in"}
for w in range(2):
    t = tr[w]; t = t[t != 0]
    tag = ((t >> 56) & 0xff).astype(int); clk = (t & ((1 << 56) - 1)).astype(np.int64)
    ev = list(zip(tag.tolist(), clk.tolist()))
    # average duration of every phase (time since the previous stamp), over the steady-state blocks
    acc = {}
    for (a, ca), (b, cb) in zip(ev[1:], ev[:-1]):
        acc.setdefault(a, []).append(ca - cb)
    print("wave 0" if w == 0 else "wave 5", {names[k].split(" [")[0]: int(np.median(v)) for k, v in sorted(acc.items())})
    b0 = [c for g, c in ev if g == 10]
    print("   block periods (arrive b0 -> arrive b0):", np.diff(b0).tolist()[:12], " total", clk[-1] - clk[0], "ticks (100 MHz s_memtime? see the ratio to the HIP-event time)")
build.build(force=True, verbose=False)

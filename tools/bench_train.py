"""Forward + backward time of the training-mode denoiser at BASELINE config 4's size (B=128 x 2048 pts, fp32):
python tools/bench_train.py [B] [N] [f32|bf16].  Prints ms per iteration and the achieved fp32 matrix throughput
(3 x 4.734 GFLOP per shape: forward + dX + dW products)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import synth, training

_pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] not in ("--streams", "--dropout")]
DROPOUT = float(sys.argv[sys.argv.index("--dropout") + 1]) if "--dropout" in sys.argv else 0.0   # e.g. 0.2 = train_chair_stage1.py as shipped
B = int(_pos[0]) if len(_pos) > 0 else 128
N = int(_pos[1]) if len(_pos) > 1 else 2048
PREC = _pos[2] if len(_pos) > 2 else "bf16"
dev = "cuda"
W = synth.make_denoiser_weights(0)
P = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in W.items()}
pc, mean, logvar, valid = synth.make_latents(B, seed=1)
seg = synth.make_seg_mask(valid, N)
var = np.exp(logvar).astype(np.float32)
idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
anc, vr = np.take_along_axis(mean, idx, axis=2), np.take_along_axis(var, idx, axis=2)
rng = np.random.default_rng(0)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x_t = cu((anc + np.sqrt(vr) * rng.standard_normal((B, 3, N))).astype(np.float32))
t = cu(rng.integers(0, 1000, size=(B,)).astype(np.int64))
args = [x_t, t, cu(pc), cu(np.concatenate([mean, var], 1).astype(np.float32)), cu(anc.transpose(0, 2, 1)), cu(vr.transpose(0, 2, 1)),
        cu(valid), cu(seg.astype(np.int32))]
noise = cu(rng.standard_normal((B, 3, N)).astype(np.float32))
opt = training.Adam(list(P.values()), lr=1e-4, max_norm=10.0)
if "--streams" in sys.argv:   # dfx_debug_train_streams: 0 = every launch of the fused path on the caller's stream, 1 = default, else a set of SS_* bits
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_train_streams(int(sys.argv[sys.argv.index("--streams") + 1]))
if "--layerwise" in sys.argv:   # the layer-by-layer feed-forward kernels (A/B against the fused ones)
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_train_fused(0)


_step = [0]


def it():
    opt.zero_grad()
    _step[0] += 1
    loss = training.masked_mse(noise, training.denoiser_train_forward(P, *args, precision=PREC, dropout=(DROPOUT, 1000 + _step[0]) if DROPOUT > 0 else None), None)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20 if "--long" in sys.argv else 5
for _ in range(K):
    loss = it()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
flops = 3 * 4.734e9 * B * (N / 2048)
peak = 157.3 if PREC == "f32" else 2500.0
print(f"training iteration (forward + backward + clip + Adam), B={B} N={N}, dropout {DROPOUT}, matrix products in {PREC}: {ms:.2f} ms = {B / ms * 1e3:.0f} shapes/s, "
      f"{flops / ms / 1e9:.1f} TFLOP/s of the {peak} TFLOP/s {PREC} matrix peak ({flops / ms / 1e9 / peak * 100:.1f} %), loss {float(loss.detach()):.4f}, "
      f"workspace {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak, optimiser in {'one launch' if opt.last_step_was_flat else 'one launch per tensor'}")
if PREC == "bf16" and "--ab" in sys.argv:   # the same loop through the layer-by-layer feed-forward kernels
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_train_fused(0)
    for _ in range(2):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        loss = it()
    torch.cuda.synchronize()
    print(f"layer-by-layer feed-forward (dfx_debug_train_fused(0)): {(time.perf_counter() - t0) / K * 1e3:.1f} ms, loss {float(loss.detach()):.4f}")
    _ffi.lib().dfx_debug_train_fused(1)

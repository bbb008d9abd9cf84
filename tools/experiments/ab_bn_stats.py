import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from difffacto_amd import _ffi, synth
from oracle import pointnet_v2_train as pt
from test_gpu_encoder_train import _run, ZERO_GRAD
for B, N in [(8, 2048), (8, 1024), (16, 1024)]:
    rng = np.random.Generator(np.random.PCG64(B * 77 + N))
    W = synth.make_pointnet_v2_weights(4)
    x = rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)
    attn = np.eye(4, dtype=np.float32)[rng.integers(0, 4, size=(B, N))]
    dm, dv = rng.standard_normal((B, 4, 256)).astype(np.float32), rng.standard_normal((B, 4, 256)).astype(np.float32)
    ref = pt.outputs_and_grads(W, x, attn, dm, dv)
    fused = _run(W, x, attn, dm, dv)
    _ffi.lib().dfx_debug_bn_fused_stats(0)
    two = _run(W, x, attn, dm, dv)
    two2 = _run(W, x, attn, dm, dv)
    _ffi.lib().dfx_debug_bn_fused_stats(1)
    def worst(a, b):
        w = (0, "")
        for k, gr in b.items():
            if k in ZERO_GRAD or k == "bn4.bias": continue
            e = np.abs(a[k] - gr).max() / max(np.abs(gr).max(), 1e-30)
            if e > w[0]: w = (e, k)
        return w
    print(B, N, "fused-oracle", worst(fused["grads"], ref["grads"]), "two-oracle", worst(two["grads"], ref["grads"]), "fused-two", worst(fused["grads"], two["grads"]),
          "two-two", worst(two2["grads"], two["grads"]), "m", np.abs(fused["m"] - ref["m"]).max(), np.abs(two["m"] - ref["m"]).max())
    for k in ("bn1.running_mean", "bn1.running_var", "bn4.running_var"):
        print("   ", k, np.abs(fused["running"][k] - ref["running"][k]).max(), np.abs(two["running"][k] - ref["running"][k]).max())

"""Timing of the pointnet2 / chamfer kernels at the sizes the reference uses (GPU box).  Prints one line per op:
average launch time (HIP events on the launch stream), algorithmic bytes moved, achieved GB/s vs the 6.3 TB/s
achievable HBM rate, and for FPS the time per selected point (the op is a dependent chain, not a stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from difffacto_amd.pointnet2_ops import pointnet2_utils as pu
from difffacto_amd.metrics import ChamferFunction


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def line(name, ms, nbytes, extra=""):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(f"{name:58s} {ms * 1e3:9.1f} us  {nbytes / 1e6:9.2f} MB  {gbs:8.1f} GB/s ({gbs / 6300 * 100:5.2f} % of 6.3 TB/s) {extra}")


torch.manual_seed(0)
dev = "cuda"
for B, N, M in [(1, 8192, 2048), (128, 8192, 2048), (128, 2048, 512), (128, 512, 128)]:
    xyz = torch.randn(B, N, 3, device=dev)
    ms = timeit(lambda: pu.furthest_point_sample(xyz, M))
    line(f"furthest_point_sample B={B} N={N} -> M={M}", ms, B * (N * 12 + M * 4), f"{ms * 1e6 / M:7.1f} ns / selected point / cloud-wave")
for B, N, M, r, ns in [(128, 2048, 512, 0.2, 64), (128, 512, 128, 0.4, 64)]:
    xyz = torch.rand(B, N, 3, device=dev) * 2 - 1
    new_xyz = xyz[:, :M].contiguous()
    ms = timeit(lambda: pu.ball_query(r, ns, xyz, new_xyz))
    line(f"ball_query B={B} N={N} M={M} r={r} ns={ns}", ms, B * ((N + M) * 12 + M * ns * 4))
    idx = pu.ball_query(r, ns, xyz, new_xyz)
    for C in (3, 128):
        feats = torch.randn(B, C, N, device=dev)
        ms = timeit(lambda: pu.grouping_operation(feats, idx))
        line(f"grouping_operation B={B} C={C} N={N} np={M} ns={ns}", ms, B * (M * ns * 4 + C * M * ns * 4 * 2))
for B, C, N, M in [(128, 3, 4, 2048), (128, 3, 8192, 2048)]:
    feats = torch.randn(B, C, N, device=dev)
    idx = torch.randint(0, N, (B, M), device=dev, dtype=torch.int32)
    ms = timeit(lambda: pu.gather_operation(feats, idx))
    line(f"gather_operation B={B} C={C} N={N} M={M}", ms, B * (M * 4 + C * M * 4 * 2))
unknown, known = torch.randn(128, 2048, 3, device=dev), torch.randn(128, 512, 3, device=dev)
ms = timeit(lambda: pu.three_nn(unknown, known))
line("three_nn B=128 n=2048 m=512", ms, 128 * ((2048 + 512) * 12 + 2048 * 24), f"{128 * 2048 * 512 / (ms * 1e-3) / 1e9:7.1f} G pair-dist/s")
a, b = torch.randn(128, 2048, 3, device=dev), torch.randn(128, 2048, 3, device=dev)
ms = timeit(lambda: ChamferFunction.apply(a, b))
line("chamfer forward B=128 N=M=2048", ms, 128 * 2 * (2048 * 12 + 2048 * 8), f"{2 * 128 * 2048 * 2048 / (ms * 1e-3) / 1e9:7.1f} G pair-dist/s")

# ---- set-abstraction layers of PointNet2SSG (python/difffacto/models/encoders/pointnet2.py:18-45), eval mode ----
import numpy as np  # noqa: E402
from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule  # noqa: E402

B = 128
xyz = torch.rand(B, 2048, 3, device=dev) * 2 - 1
feats = torch.randn(B, 4, 2048, device=dev)
specs = [dict(npoint=512, radius=0.2, nsample=64, mlp=[4, 64, 64, 128]), dict(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256]),
         dict(mlp=[256, 256, 512, 1024])]
with torch.no_grad():
    for k, sp in enumerate(specs):
        mod = PointnetSAModule(**{**sp, "mlp": list(sp["mlp"])}).to(dev).eval()
        mlp = [sp["mlp"][0] + 3] + sp["mlp"][1:]
        new_xyz = mod._centres(xyz)
        M = 1 if new_xyz is None else new_xyz.shape[1]
        ns = xyz.shape[1] if new_xyz is None else sp["nsample"]
        rows = B * M * ns
        flops = 2 * rows * sum(a * b for a, b in zip(mlp[:-1], mlp[1:]))
        for name, fg in (("fused" if k < 2 else "general", False), ("general", True)):
            if k == 2 and not fg:
                continue
            ms = timeit(lambda: mod._forward_native(0, xyz, new_xyz, feats, force_general=fg), iters=10)
            print(f"SA{k + 1} group+MLP{mlp}+maxpool ({name:7s}) B={B} M={M} ns={ns}: {ms * 1e3:9.1f} us  {flops / (ms * 1e-3) / 1e12:6.1f} TFLOP/s fp32 "
                  f"({flops / (ms * 1e-3) / 1e12 / 157.3 * 100:4.1f} % of the 157.3 TFLOP/s fp32 matrix peak); grouped tensor that never exists: {rows * mlp[0] * 4 / 1e6:.0f} MB")
        ms = timeit(lambda: mod(xyz, feats), iters=5)
        print(f"SA{k + 1} whole module (FPS + gather + ball query + fused MLP/pool): {ms * 1e3:9.1f} us")
        xyz, feats = mod(xyz, feats)

# ---- approximate EMD at the evaluation setting (datasets/evaluation_utils.py:84-89: EMD(0.002, 10000), batches of 32 x 2048) ----
from difffacto_amd.metrics import EMD  # noqa: E402
a, b = torch.rand(32, 2048, 3, device=dev), torch.rand(32, 2048, 3, device=dev)
emd = EMD(0.002, 10000, True)
ms = timeit(lambda: emd(a, b), iters=3, warmup=1)
print(f"EMD auction (eps 0.002, up to 10000 iterations, one persistent workgroup per pair) B=32 n=2048: {ms:9.1f} ms  "
      f"(the reference issues 70 000 kernel launches for the same call)")

# ---- PointNetV2 part encoder (python/difffacto/models/encoders/pointnet.py:187-213), eval mode, B = 128 x 2048 points ----
from difffacto_amd.encoders import PointNetV2  # noqa: E402
enc = PointNetV2(zdim=256, point_dim=3, per_part_mlp=True, num_anchors=4).to(dev).eval()
x = torch.rand(128, 2048, 3, device=dev) * 2 - 1
attn = torch.eye(4, device=dev)[torch.randint(0, 4, (128, 2048), device=dev)]
with torch.no_grad():
    ms = timeit(lambda: enc(x, attn), iters=10)
fl = 2 * 128 * 2048 * (3 * 128 + 128 * 128 + 128 * 256 + 256 * 512) + 2 * 2 * 128 * 4 * (512 * 256 + 256 * 128 + 128 * 256)
print(f"PointNetV2 forward (trunk rows = 262144, masked max-pool, grouped heads): {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s fp32")
# a whole chip of auctions: the evaluator (difffacto_amd/evaluation.py) sends the all-pairs matrices out PAIRS_PER_LAUNCH pairs at a time
from difffacto_amd.evaluation import PAIRS_PER_LAUNCH  # noqa: E402
a, b = torch.rand(PAIRS_PER_LAUNCH, 2048, 3, device=dev), torch.rand(PAIRS_PER_LAUNCH, 2048, 3, device=dev)
ms = timeit(lambda: emd(a, b), iters=2, warmup=1)
print(f"EMD auction, {PAIRS_PER_LAUNCH} pairs per launch (one workgroup per pair, one per compute unit): {ms:9.1f} ms = {PAIRS_PER_LAUNCH / ms * 1e3:7.0f} pairs/s "
      f"(32 pairs per launch as the reference batches them: see the line above)")

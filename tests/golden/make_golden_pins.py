"""Dev-container-only: reference pins for the oracles that round 1 left "parity unpinned".

    python tests/golden/make_golden_pins.py

Imports the reference's own pure-torch helpers (never shipped to the GPU box) and stores inputs + outputs:

  fps_ref_*.npz        `farthest_point_sample` (python/difffacto/models/encoders/pointnet2_utils.py:60-81) with
                       `torch.randint` patched to return 0, so that it starts at index 0 like the CUDA kernel
                       (SRC/sampling_gpu.cu:86-87).  The clouds avoid the two places where that helper and the kernel
                       differ by construction: no point with |p|^2 <= 1e-3 (the kernel's origin skip, :100-101) and no
                       near-ties — every selection's winner leads the runner-up by a relative margin > 4e-6 (1e-6 for the 8192-point cloud) in float64,
                       8-30 times the rounding difference between torch's sum of squares and the kernel's fma chain
                       (<= 2 ulp = 1.2e-7 relative), and the fp32 reference is checked to reproduce the float64 selection.
  chamfer_ref_*.npz    `distChamfer` (python/difffacto/datasets/evaluation_utils.py:93-103): expanded-form squared
                       distances, so it pins chamfer.cu's direct-difference form only to a tolerance (stated in the test).
  three_nn_ref_*.npz   3-NN indices / squared distances and the inverse-distance interpolation through the reference's pure-torch
                       feature-propagation path (pointnet2_utils.py:289-299, index_points :41-57).
  pn2_torch_ballquery_*.npz  three more `query_ball_point` cases (pointnet2_utils.py:84-104): centres with no point in
                       range, nsample overflow (more hits than nsample), nsample > hits (padding), N not a multiple of 64.
                       The helper excludes d2 > r^2, the kernel includes d2 < r^2: identical unless d2 == r^2 exactly,
                       which the script checks does not occur.  A centre without any hit gives index N in the helper
                       (out of range) and 0 in the kernel (`torch::zeros` output, SRC/ball_query.cpp:19-21): stored as -1
                       and compared as "kernel row is all zeros".
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402

F32 = np.float32


def fps_f64_margin(xyz, npoint):
    """float64 FPS from index 0; returns the index list and the smallest relative lead of a winner over the runner-up."""
    x = xyz.astype(np.float64)
    d = np.full(len(x), 1e10)
    cur, out, margin = 0, [0], np.inf
    for _ in range(1, npoint):
        d = np.minimum(d, ((x - x[cur]) ** 2).sum(1))
        cur = int(np.argmax(d))
        top = d[cur]
        d[cur] = -1.0
        margin = min(margin, (top - d.max()) / top)
        d[cur] = top
        out.append(cur)
    return out, margin


def gen_fps():
    ref_import.import_reference()
    from difffacto.models.encoders import pointnet2_utils as rpu
    real_randint = torch.randint
    for tag, B, N, M, seed in (("N512_M128", 2, 512, 128, 101), ("N2048_M512", 2, 2048, 512, 102), ("N8192_M2048", 1, 8192, 2048, 103)):
        rng = np.random.Generator(np.random.PCG64(seed))
        clouds = []
        for b in range(B):
            for attempt in range(200):   # redraw until the float64 selection has a comfortable margin everywhere
                xyz = rng.uniform(0.5, 1.5, size=(N, 3)).astype(F32)   # away from the origin: no |p|^2 <= 1e-3
                idx64, margin = fps_f64_margin(xyz, M)
                if margin > (4e-6 if N < 8192 else 1e-6):
                    break
            else:
                raise RuntimeError("no tie-free cloud found")
            clouds.append(xyz)
            print(f"  fps {tag} cloud {b}: margin {margin:.2e} after {attempt + 1} draws")
        xyz = np.stack(clouds)
        torch.randint = lambda *a, **k: torch.zeros(a[2] if len(a) > 2 else k.get("size"), dtype=torch.long)
        try:
            idx = rpu.farthest_point_sample(torch.from_numpy(xyz), M).numpy()
        finally:
            torch.randint = real_randint
        for b in range(B):
            assert idx[b].tolist() == fps_f64_margin(xyz[b], M)[0], "fp32 reference disagrees with float64: near-tie"
        np.savez_compressed(os.path.join(HERE, f"fps_ref_{tag}.npz"), xyz=xyz, npoint=np.array(M), idx=idx.astype(np.int32))
        print("wrote fps_ref_" + tag, idx.shape)


def gen_chamfer():
    ref_import.import_reference()
    from difffacto.datasets.evaluation_utils import distChamfer
    for tag, B, N, seed in (("B3_N256", 3, 256, 201), ("B2_N2048", 2, 2048, 202)):
        rng = np.random.Generator(np.random.PCG64(seed))
        a = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
        b = (a[:, rng.permutation(N)] + 0.05 * rng.standard_normal((B, N, 3))).astype(F32)
        # distChamfer(a, b): P[i, j] = |a_i|^2 + |b_j|^2 - 2 a_i.b_j ; returns (P.min(1), P.min(2)) = (per b_j, per a_i)
        d_b, d_a = distChamfer(torch.from_numpy(a), torch.from_numpy(b))
        np.savez_compressed(os.path.join(HERE, f"chamfer_ref_{tag}.npz"), a=a, b=b, dist_a=d_a.numpy().astype(F32),
                            dist_b=d_b.numpy().astype(F32))
        print("wrote chamfer_ref_" + tag)


def gen_ballquery():
    ref_import.import_reference()
    from difffacto.models.encoders.pointnet2_utils import query_ball_point
    cases = (("nohit", 2, 300, 40, 8, 0.12, 301),       # sparse cloud, small radius: many centres have no neighbour but themselves / none
             ("overflow", 2, 1000, 64, 16, 0.6, 302),   # far more hits than nsample everywhere
             ("pad", 1, 130, 33, 64, 0.3, 303))         # nsample > hits: padding with the first hit; ragged sizes
    for tag, B, N, M, ns, r, seed in cases:
        rng = np.random.Generator(np.random.PCG64(seed))
        xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
        new_xyz = xyz[:, rng.permutation(N)[:M]].copy()
        if tag == "nohit":   # half of the centres far outside the cloud: no hit at all
            new_xyz[:, ::2] += 5.0
        d2 = ((new_xyz[:, :, None].astype(np.float64) - xyz[:, None].astype(np.float64)) ** 2).sum(-1)
        assert np.abs(d2 - float(F32(r)) ** 2).min() > 1e-6, "a distance sits on the radius: '<' vs '<=' would matter"
        idx = query_ball_point(r, ns, torch.from_numpy(xyz), torch.from_numpy(new_xyz)).numpy().astype(np.int64)
        nohit = (idx == N).all(-1)
        idx[idx == N] = -1
        np.savez_compressed(os.path.join(HERE, f"pn2_torch_ballquery_{tag}.npz"), xyz=xyz, new_xyz=new_xyz, radius=np.array(r, F32),
                            nsample=np.array(ns), idx=idx)
        print("wrote pn2_torch_ballquery_" + tag, "centres without a hit:", int(nohit.sum()), "of", nohit.size)


def gen_three_nn():
    """3-NN + inverse-distance interpolation through the reference's own pure-torch feature-propagation code path
    (models/encoders/pointnet2_utils.py:289-299: square_distance -> sort -> first three; index_points + weighted sum).  Pins the
    three_nn kernel's neighbour choice and order (interpolate_gpu.cu:13-59: ascending distance) and three_interpolate's
    weighted gather (:72-101) on clouds whose four nearest distances are separated by > 1e-4 relative (no ties to break)."""
    ref_import.import_reference()
    from difffacto.models.encoders.pointnet2_utils import index_points, square_distance
    for tag, B, n, m, c, seed in (("n300_m64_c5", 2, 300, 64, 5, 401), ("n64_m700_c3", 1, 64, 700, 3, 402)):
        rng = np.random.Generator(np.random.PCG64(seed))
        for attempt in range(100):
            unknown = rng.uniform(-1, 1, size=(B, n, 3)).astype(F32)
            known = rng.uniform(-1, 1, size=(B, m, 3)).astype(F32)
            d64 = np.sort(((unknown[:, :, None].astype(np.float64) - known[:, None].astype(np.float64)) ** 2).sum(-1), axis=-1)[:, :, :4]
            if ((d64[:, :, 1:] - d64[:, :, :-1]) / d64[:, :, 1:]).min() > 1e-4:
                break
        else:
            raise RuntimeError("no tie-free case found")
        d, idx = square_distance(torch.from_numpy(unknown), torch.from_numpy(known)).sort(dim=-1)
        d, idx = d[:, :, :3], idx[:, :, :3]
        feats = rng.standard_normal((B, c, m)).astype(F32)
        recip = 1.0 / (torch.sqrt(d.clamp_min(0)) + 1e-8)          # PointnetFPModule's weights (pointnet2_modules.py:196-199)
        weight = (recip / recip.sum(2, keepdim=True)).numpy().astype(F32)
        interp = torch.sum(index_points(torch.from_numpy(feats).permute(0, 2, 1), idx) * torch.from_numpy(weight)[..., None], dim=2)
        np.savez_compressed(os.path.join(HERE, f"three_nn_ref_{tag}.npz"), unknown=unknown, known=known, idx=idx.numpy().astype(np.int32),
                            dist2=d.numpy().astype(F32), feats=feats, weight=weight, interp=interp.permute(0, 2, 1).numpy().astype(F32))
        print("wrote three_nn_ref_" + tag)


def main():
    gen_three_nn()
    gen_fps()
    gen_chamfer()
    gen_ballquery()


if __name__ == "__main__":
    main()

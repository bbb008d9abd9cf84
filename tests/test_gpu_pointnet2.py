"""GPU parity: libdfx pointnet2 kernels (through the C-ABI / drop-in python API) vs the C oracle.
Bit-exact for indices and gathers; atomics-based gradients within fp32 summation-order tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pointnet2 as opn  # noqa: E402


@pytest.fixture(scope="module")
def pu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.pointnet2_ops import pointnet2_utils
    return pointnet2_utils


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cloud(rng, B, N, dup=0.0, origin=0):
    xyz = rng.standard_normal((B, N, 3)).astype(np.float32)
    if dup > 0:   # duplicated points are common in the real data (np.random.choice(replace=True))
        k = int(N * dup)
        for b in range(B):
            src = rng.integers(0, N, size=k)
            dst = rng.integers(1, N, size=k)
            xyz[b, dst] = xyz[b, src]
    for b in range(B):
        for _ in range(origin):
            xyz[b, rng.integers(1, N)] = rng.uniform(-0.01, 0.01, size=3)
    return xyz


@pytest.mark.parametrize("B,N,M", [(2, 2048, 512), (3, 512, 128), (1, 8192, 2048), (2, 700, 64), (2, 5000, 300),
                                   (1, 12000, 128), (1, 20000, 64), (4, 33, 33), (2, 1, 1), (1, 3, 3)])
def test_fps_matches_oracle(pu, B, N, M):
    rng = np.random.default_rng(N + M)
    xyz = cloud(rng, B, N, dup=0.2 if N > 100 else 0.0, origin=3 if N > 100 else 0)
    got = pu.furthest_point_sample(dev(xyz), M).cpu().numpy()
    ref = opn.furthest_point_sampling(xyz, M)
    assert got.dtype == np.int32 and got.shape == (B, M)
    assert np.array_equal(got, ref), np.argwhere(got != ref)[:5]
    # coordinates too (duplicates make index-only checks fragile)
    assert np.array_equal(np.take_along_axis(xyz, got[..., None].astype(np.int64), 1),
                          np.take_along_axis(xyz, ref[..., None].astype(np.int64), 1))


def test_fps_tie_and_skip_edge_cases(pu):
    n = 600
    pts = np.zeros((1, n, 3), np.float32)
    pts[0, :, 0] = 1.0
    pts[0, 1] = pts[0, 2] = [-3.0, 0, 0]
    assert pu.furthest_point_sample(dev(pts), 3).cpu().numpy().tolist() == opn.furthest_point_sampling(pts, 3).tolist() == [[0, 2, 0]]
    z = np.zeros((2, 40, 3), np.float32)
    assert pu.furthest_point_sample(dev(z), 5).cpu().numpy().tolist() == [[0] * 5] * 2
    assert pu.furthest_point_sample(dev(np.ones((1, 5, 3), np.float32)), 0).shape == (1, 0)


def test_fps_properties_full_size(pu):
    """BASELINE config 3 size (gen_car: 8192 -> 2048): size-independent properties."""
    rng = np.random.default_rng(5)
    xyz = rng.standard_normal((4, 8192, 3)).astype(np.float32)
    idx = pu.furthest_point_sample(dev(xyz), 2048).cpu().numpy()
    for b in range(4):
        assert len(set(idx[b].tolist())) == 2048            # no repeats on distinct points
        sel = xyz[b, idx[b]]
        # greedy max-min: the distance of pick j to the earlier picks is non-increasing in j
        d = np.array([np.min(((sel[:j] - sel[j]) ** 2).sum(1)) for j in range(1, 200)])
        assert np.all(d[1:] <= d[:-1] + 1e-6)


@pytest.mark.parametrize("B,N,M,r,ns", [(2, 2048, 512, 0.2, 64), (2, 512, 128, 0.4, 64), (1, 8192, 256, 0.1, 32),
                                        (1, 13000, 64, 0.3, 16), (3, 100, 7, 0.5, 5), (1, 70, 70, 10.0, 128)])
def test_ball_query_matches_oracle(pu, B, N, M, r, ns):
    rng = np.random.default_rng(N * 7 + M)
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)
    new_xyz = xyz[:, rng.permutation(N)[:M]].copy()
    new_xyz[:, -1] = 50.0    # a centre with no neighbour -> zeros
    got = pu.ball_query(r, ns, dev(xyz), dev(new_xyz)).cpu().numpy()
    ref = opn.ball_query(r, ns, xyz, new_xyz)
    assert np.array_equal(got, ref)
    assert np.all(got[:, -1] == 0)


@pytest.mark.parametrize("B,C,N,M", [(2, 3, 4, 2048), (4, 1, 4, 2048), (1, 3, 8192, 2048), (3, 131, 512, 128), (2, 7, 1, 5)])
def test_gather_matches_oracle(pu, B, C, N, M):
    rng = np.random.default_rng(C + N)
    pts = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, M)).astype(np.int32)
    t = dev(pts).requires_grad_(True)
    out = pu.gather_operation(t, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), opn.gather_points(pts, idx))
    g = rng.standard_normal((B, C, M)).astype(np.float32)
    out.backward(dev(g))
    assert np.allclose(t.grad.cpu().numpy(), opn.gather_points_grad(g, idx, N), atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("B,C,N,NP,NS", [(2, 7, 2048, 512, 64), (2, 131, 512, 128, 64), (1, 3, 100, 9, 5)])
def test_group_matches_oracle(pu, B, C, N, NP, NS):
    rng = np.random.default_rng(C * NP)
    pts = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, NP, NS)).astype(np.int32)
    t = dev(pts).requires_grad_(True)
    out = pu.grouping_operation(t, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), opn.group_points(pts, idx))
    g = rng.standard_normal((B, C, NP, NS)).astype(np.float32)
    out.backward(dev(g))
    assert np.allclose(t.grad.cpu().numpy(), opn.group_points_grad(g, idx, N), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("B,n,m,c", [(2, 512, 128, 16), (1, 2048, 512, 8), (2, 37, 5000, 3), (1, 5, 3, 2)])
def test_three_nn_interpolate_match_oracle(pu, B, n, m, c):
    rng = np.random.default_rng(n + m)
    unknown = rng.standard_normal((B, n, 3)).astype(np.float32)
    known = rng.standard_normal((B, m, 3)).astype(np.float32)
    known[:, 1] = known[:, 0]   # exact duplicate -> strict '<' keeps the earlier index
    dist, idx = pu.three_nn(dev(unknown), dev(known))
    rd, ri = opn.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(dist.cpu().numpy(), rd)
    w = rng.random((B, n, 3)).astype(np.float32)
    feats = rng.standard_normal((B, c, m)).astype(np.float32)
    t = dev(feats).requires_grad_(True)
    out = pu.three_interpolate(t, idx, dev(w))
    assert np.array_equal(out.detach().cpu().numpy(), opn.three_interpolate(feats, ri, w))
    g = rng.standard_normal((B, c, n)).astype(np.float32)
    out.backward(dev(g))
    assert np.allclose(t.grad.cpu().numpy(), opn.three_interpolate_grad(g, ri, w, m), atol=1e-3, rtol=1e-4)


def test_query_and_group_and_sa_module(pu):
    from difffacto_amd.pointnet2_ops import pointnet2_modules as pm
    rng = np.random.default_rng(1)
    B, N = 2, 1024
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)
    feats = rng.standard_normal((B, 4, N)).astype(np.float32)
    sa = pm.PointnetSAModule(npoint=128, radius=0.3, nsample=32, mlp=[4, 16, 32], use_xyz=True).cuda().eval()
    new_xyz, new_f = sa(dev(xyz), dev(feats))
    assert new_xyz.shape == (B, 128, 3) and new_f.shape == (B, 32, 128)
    # the same pipeline with the oracle's indices + the module's own torch MLP
    fi = opn.furthest_point_sampling(xyz, 128)
    centres = np.take_along_axis(xyz, fi[..., None].astype(np.int64), 1)
    assert np.array_equal(new_xyz.cpu().numpy(), centres)
    bi = opn.ball_query(0.3, 32, xyz, centres)
    gx = opn.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), bi) - centres.transpose(0, 2, 1)[..., None]
    gf = opn.group_points(feats, bi)
    with torch.no_grad():
        ref = sa.mlps[0](dev(np.concatenate([gx, gf], 1))).amax(3)
    assert torch.allclose(new_f, ref, atol=1e-5)


def test_dtype_and_contiguity_errors(pu):
    x = torch.zeros(1, 8, 3, device="cuda")
    with pytest.raises(RuntimeError, match="contiguous"):
        pu.furthest_point_sample(torch.zeros(1, 3, 8, device="cuda").transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="int"):
        pu.gather_operation(torch.zeros(1, 3, 8, device="cuda"), torch.zeros(1, 2, dtype=torch.int64, device="cuda"))
    with pytest.raises(RuntimeError, match="float"):
        pu.furthest_point_sample(x.double(), 4)


@pytest.mark.parametrize("B,N,M", [(3, 2048, 2048), (2, 500, 3000), (1, 1, 7)])
def test_chamfer_matches_oracle(pu, B, N, M):
    from difffacto_amd.metrics import ChamferFunction, ChamferDistanceL2
    rng = np.random.default_rng(N + M)
    a = rng.standard_normal((B, N, 3)).astype(np.float32)
    b = rng.standard_normal((B, M, 3)).astype(np.float32)
    if M > 2:
        b[:, 1] = b[:, 0]    # duplicate: strict '<' keeps the first index
    ta, tb = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    d1, d2 = ChamferFunction.apply(ta, tb)
    r1, r2, i1, i2 = opn.chamfer_forward(a, b)
    assert np.array_equal(d1.detach().cpu().numpy(), r1) and np.array_equal(d2.detach().cpu().numpy(), r2)
    g1 = rng.standard_normal((B, N)).astype(np.float32)
    g2 = rng.standard_normal((B, M)).astype(np.float32)
    (d1 * dev(g1)).sum().add((d2 * dev(g2)).sum()).backward()
    gx1, gx2 = opn.chamfer_backward(a, b, i1, i2, g1, g2)
    assert np.allclose(ta.grad.cpu().numpy(), gx1, atol=1e-4, rtol=1e-4)
    assert np.allclose(tb.grad.cpu().numpy(), gx2, atol=2e-3, rtol=1e-3)
    # brute-force cross-check of the value (distChamfer-style, evaluation_utils.py:93-103)
    full = ((a[:, :, None] - b[:, None]) ** 2).sum(-1)
    assert np.allclose(r1, full.min(2), atol=1e-5) and np.allclose(r2, full.min(1), atol=1e-5)
    cd = ChamferDistanceL2()(dev(a), dev(b)).item()
    assert abs(cd - (full.min(2).mean() + full.min(1).mean())) < 1e-5

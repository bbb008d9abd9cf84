"""GPU parity of the auction EMD through the C-ABI vs the C oracle (oracle/pointnet2.c: a sequential
restatement of emd_cuda.cu): assignment and squared distances bit-exact — integer work and correctly-rounded fp32 /
fp64 scalar arithmetic in the same order.  The oracle itself is unpinned against the reference (its CUDA source cannot be
built here); its tie rule for GetMax's data race (largest bidder index) is shared by both sides."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(B, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(0, 1, (B, n, 3)).astype(np.float32), rng.uniform(0, 1, (B, n, 3)).astype(np.float32)


@pytest.mark.parametrize("B,n,iters", [(3, 1024, 300), (2, 2048, 60), (2, 1000, 200), (1, 64, 5000), (2, 8192, 3)])
def test_emd_forward_bit_exact_vs_oracle(B, n, iters):
    from difffacto_amd.metrics import emdFunction
    from oracle import pointnet2 as o
    a, b = _clouds(B, n, n + iters)
    d_ref, as_ref = o.emd_forward(a, b, 0.002, iters)
    d, asg = emdFunction.apply(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 0.002, iters)
    assert np.array_equal(asg.cpu().numpy(), as_ref)
    assert np.array_equal(d.cpu().numpy(), d_ref)


def test_emd_converged_is_a_permutation_and_near_optimal():
    """Size-independent property: once the auction has converged (no unassigned point left before the last iteration) the
    assignment is a permutation and its cost is within n * eps of the optimal transport cost."""
    from difffacto_amd.metrics import EMD
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial.distance import cdist
    a, b = _clouds(2, 512, 7)
    dist, asg = EMD(0.002, 20000)(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    cost = EMD(0.002, 20000, dist_only=True)(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    for i in range(2):
        assert sorted(asg[i].cpu().tolist()) == list(range(512))
        C = cdist(a[i], b[i])
        r, c = linear_sum_assignment(C)
        assert C[r, c].mean() - 1e-6 <= cost[i] <= C[r, c].mean() + 0.002 + 1e-6


def test_emd_backward_vs_oracle():
    from difffacto_amd.metrics import emdFunction
    from oracle import pointnet2 as o
    a, b = _clouds(2, 256, 3)
    x1 = torch.from_numpy(a).cuda().requires_grad_(True)
    x2 = torch.from_numpy(b).cuda().requires_grad_(True)
    d, asg = emdFunction.apply(x1, x2, 0.005, 50)
    w = torch.linspace(0.5, 1.5, 256, device="cuda")[None].expand(2, -1)
    (d * w).sum().backward()
    ref = o.emd_backward(a, b, w.cpu().numpy(), asg.cpu().numpy())
    assert np.abs(x1.grad.cpu().numpy() - ref).max() < 1e-6
    assert float(x2.grad.abs().max()) == 0.0


@pytest.mark.parametrize("state_global", [0, 1])
def test_emd_tail_iterations_and_ties_bit_exact(state_global):
    """The auction's tail (<= 8 unassigned points: a separate path of the kernel — eight wavefronts share the scans, one finishes the
    iteration in registers) stopped at many different iterations, so that the last, non-evicting iteration (:209-211) falls on 1, 2,
    .. bidders as well as on many; ragged n; and lattice clouds: exact ties between targets (first index wins), equal increments on
    one target (largest bidder index wins) and coincident points (distance 0).  With the state in LDS and in the workspace."""
    from difffacto_amd import _ffi
    from difffacto_amd.metrics import emdFunction
    from oracle import pointnet2 as o
    rng = np.random.Generator(np.random.PCG64(11))
    cases = []
    for n in (257, 96):
        a, b = _clouds(1, n, n)
        cases += [(a, b, it) for it in (1, 2, 3, 17, 40, 80, 120, 160, 200, 250, 300, 400, 600, 900)]
    lat = (rng.integers(0, 5, (2, 200, 3)) / 4).astype(np.float32)
    cases += [(lat[:1], lat[1:], it) for it in (5, 50, 500, 3000)] + [(lat[:1], lat[:1].copy(), 50)]
    # squared distances in (0, 2^-96) (the kernel's square root takes its library path for such a scan), alone and mixed with ordinary ones
    a, b = _clouds(1, 130, 3)
    mixed_a, mixed_b = a.copy(), b.copy()
    mixed_a[0, :40] *= np.float32(1e-17)
    mixed_b[0, 20:70] *= np.float32(1e-17)
    cases += [(a * np.float32(1e-16), b * np.float32(1e-16), 30), (mixed_a, mixed_b, 40), (mixed_a, mixed_b, 700)]
    _ffi.lib().dfx_debug_emd_state_global(state_global)
    try:
        tails = 0
        for a, b, it in cases:
            d_ref, as_ref = o.emd_forward(a, b, 0.002, it)
            d, asg = emdFunction.apply(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 0.002, it)
            assert np.array_equal(asg.cpu().numpy(), as_ref), (a.shape, it)
            assert np.array_equal(d.cpu().numpy(), d_ref), (a.shape, it)
            tails += 0 < int(o.emd_unassigned_per_iteration(a, b, 0.002, it)[-1]) <= 8
        assert tails >= 6   # the sweep does stop inside the tail (the last iteration has 1..8 bidders)
    finally:
        _ffi.lib().dfx_debug_emd_state_global(0)

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

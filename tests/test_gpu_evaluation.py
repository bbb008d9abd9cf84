"""GPU: the generation metrics (MMD / COV / 1-NNA under CD and EMD, python/difffacto/datasets/evaluation_utils.py) on the
native Chamfer / EMD kernels vs a brute-force numpy evaluation with the C oracle's EMD."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _brute(sample, ref):
    from oracle import pointnet2 as o
    cd = np.zeros((len(sample), len(ref)), np.float32)
    emd = np.zeros_like(cd)
    for i, a in enumerate(sample):
        for j, b in enumerate(ref):
            d = ((a[:, None] - b[None]) ** 2).sum(-1)
            cd[i, j] = d.min(1).mean() + d.min(0).mean()
            dist, _ = o.emd_forward(a[None], b[None], 0.002, 10000)
            emd[i, j] = np.sqrt(dist).mean()
    return cd, emd


def test_pairwise_and_all_metrics_vs_bruteforce():
    from difffacto_amd import evaluation as ev
    rng = np.random.Generator(np.random.PCG64(2))
    ref = rng.uniform(0, 1, (4, 128, 3)).astype(np.float32)
    smp = np.concatenate([ref[:2] + 0.01 * rng.standard_normal((2, 128, 3)).astype(np.float32), rng.uniform(0, 1, (3, 128, 3)).astype(np.float32)])
    smp = np.clip(smp, 0, 1)
    S, R = torch.from_numpy(smp).cuda(), torch.from_numpy(ref).cuda()
    cd, emd = ev._pairwise_EMD_CD_(S, R, batch_size=3)
    cd_ref, emd_ref = _brute(smp, ref)
    assert np.abs(cd.cpu().numpy() - cd_ref).max() < 1e-6
    assert np.abs(emd.cpu().numpy() - emd_ref).max() < 1e-6
    paired = ev.EMD_CD(S[:4], R, batch_size=3, reduced=False)
    assert np.abs(paired["MMD-CD"].cpu().numpy() - np.diag(cd_ref[:4])).max() < 1e-6
    assert np.abs(paired["MMD-EMD"].cpu().numpy() - np.diag(emd_ref[:4])).max() < 1e-6
    res = ev.compute_all_metrics(S, R, batch_size=3)
    # the two perturbed copies are the closest samples of references 0 and 1
    assert abs(float(res["lgan_mmd-CD"]) - cd_ref.min(0).mean()) < 1e-6
    emd_rs = _brute(ref, smp)[1]   # the auction is not symmetric: compute_all_metrics bids references against samples (:504)
    assert abs(float(res["lgan_mmd_smp-EMD"]) - emd_rs.T.min(1).mean()) < 1e-6
    assert float(res["lgan_cov-CD"]) == len(set(cd_ref.argmin(0).tolist())) / 4
    for k in ("1-NN-CD-acc", "1-NN-EMD-acc", "1-NN-CD-acc_t", "1-NN-CD-acc_f"):
        assert 0.0 <= float(res[k]) <= 1.0
    # 1-NNA on two identical sets: every cloud's nearest neighbour (leave-one-out) is its twin in the other set -> accuracy 0
    res_same = ev.compute_all_metrics(R.clone(), R, batch_size=4)
    assert float(res_same["1-NN-CD-acc"]) == 0.0 and float(res_same["lgan_cov-CD"]) == 1.0 and float(res_same["lgan_mmd-CD"]) == 0.0


def test_pairwise_launch_size_does_not_change_the_matrices(monkeypatch):
    """The all-pairs matrices go to the kernels PAIRS_PER_LAUNCH pairs at a time: any launch size gives the same numbers (masks too)."""
    from difffacto_amd import evaluation as ev
    g = torch.Generator(device="cuda").manual_seed(5)
    S, R = torch.rand(5, 96, 3, device="cuda", generator=g), torch.rand(7, 96, 3, device="cuda", generator=g)
    ms, mr = (torch.rand(5, 96, device="cuda", generator=g) > 0.3).float(), (torch.rand(7, 96, device="cuda", generator=g) > 0.3).float()
    whole = ev._pairwise_EMD_CD_(S, R, batch_size=32, mask_sample=ms, mask_ref=mr)
    monkeypatch.setattr(ev, "PAIRS_PER_LAUNCH", 1)
    for bs in (1, 4, 9):
        part = ev._pairwise_EMD_CD_(S, R, batch_size=bs, mask_sample=ms, mask_ref=mr)
        assert torch.equal(part[1], whole[1])                       # the auction: bit-identical per pair
        assert torch.allclose(part[0], whole[0], rtol=1e-6, atol=0)   # Chamfer means: torch reductions
    paired = ev.EMD_CD(S, R[:5], batch_size=2, reduced=False)
    assert torch.equal(paired["MMD-EMD"], torch.diagonal(ev._pairwise_EMD_CD_(S, R[:5], batch_size=2)[1]))

"""Generation metrics on libdfx's Chamfer / EMD kernels — mirrors the metric functions of
python/difffacto/datasets/evaluation_utils.py (``emd_approx`` :84-89, ``EMD_CD`` :106-140, ``_pairwise_EMD_CD_`` :143-202,
``knn`` :207-248, ``lgan_mmd_cov`` :251-278, ``compute_all_metrics`` :500-560): MMD / COV / 1-NNA under CD and EMD.

The distance work (nearest-neighbour scans, the auction) is native; what remains in torch is bookkeeping on the small
(N_sample x N_ref) distance matrices.  Point clouds are (num_clouds, n, 3) float32 device tensors."""
import torch

from .metrics import EMD, ChamferDistanceL2_split


def distChamferCUDA(x, y):
    """(dist1 (B,n), dist2 (B,m)) squared nearest-neighbour distances (evaluation_utils.py:17-19)."""
    return ChamferDistanceL2_split(reduce=False)(x, y)


def emd_approx(sample, ref):
    assert sample.size(1) == ref.size(1), "Not sure what would EMD do in this case"
    return EMD(0.002, 10000, True)(sample, ref)   # (B,)


# One workgroup of the auction kernel serves one pair of clouds and owns a compute unit while it runs (its state fills most of the
# 160 KiB of LDS): a launch of the reference's `batch_size` (32) pairs would leave 7/8 of the chip idle for the 54 ms an auction
# takes.  The pairs are independent, so they go to the kernels this many at a time whatever `batch_size` says (never fewer than it).
PAIRS_PER_LAUNCH = 1024


def EMD_CD(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, reduced=True):
    """Paired CD / EMD of sample i vs reference i."""
    assert sample_pcs.shape[0] == ref_pcs.shape[0]
    cd, emd = [], []
    step = max(int(batch_size), PAIRS_PER_LAUNCH)
    for s in range(0, sample_pcs.shape[0], step):
        a, b = sample_pcs[s:s + step].contiguous(), ref_pcs[s:s + step].contiguous()
        dl, dr = distChamferCUDA(a, b)
        cd.append(dl.mean(1) + dr.mean(1))
        emd.append(emd_approx(a, b))
    cd, emd = torch.cat(cd), torch.cat(emd)
    return {"MMD-CD": cd.mean() if reduced else cd, "MMD-EMD": emd.mean() if reduced else emd}


def _pairwise_EMD_CD_(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, verbose=False, mask_sample=None, mask_ref=None):
    """All-pairs (N_sample, N_ref) CD and EMD matrices; optional per-point masks weight the two Chamfer directions.
    (The reference walks sample by sample, `batch_size` references per call; here pair p = (p // N_ref, p % N_ref) of the flattened
    matrix goes out in launches of PAIRS_PER_LAUNCH — the same kernels on the same pairs, a full chip per launch.)"""
    Ns, Nr = sample_pcs.shape[0], ref_pcs.shape[0]
    step = max(int(batch_size), PAIRS_PER_LAUNCH)
    all_cd, all_emd = [], []
    for p0 in range(0, Ns * Nr, step):
        p = torch.arange(p0, min(Ns * Nr, p0 + step), device=sample_pcs.device)
        i, r = torch.div(p, Nr, rounding_mode="floor"), p % Nr
        smp, ref = sample_pcs[i].contiguous(), ref_pcs[r].contiguous()
        dl, dr = distChamferCUDA(smp, ref)
        dl_mean = dl.mean(1) if mask_sample is None else (dl * mask_sample[i]).sum(1) / mask_sample[i].sum(1)
        dr_mean = dr.mean(1) if mask_ref is None else (dr * mask_ref[r]).sum(1) / mask_ref[r].sum(1)
        all_cd.append(dl_mean + dr_mean)
        all_emd.append(emd_approx(smp, ref))
    return torch.cat(all_cd).view(Ns, Nr), torch.cat(all_emd).view(Ns, Nr)


def knn(Mxx, Mxy, Myy, k, sqrt=False, one_way=False):
    """Leave-one-out k-NN two-sample test (1-NNA for k = 1) on the block matrix [[Mxx, Mxy], [Mxy^T, Myy]]: a cloud is
    predicted to belong to set x when at least k/2 of its k nearest other clouds do.  Returns the reference's dict
    (tp, fp, fn, tn, precision, recall, acc_t, acc_f, acc) as 0-dim tensors."""
    nx, ny = Mxx.size(0), Myy.size(0)
    is_x = torch.cat((torch.ones(nx), torch.zeros(ny))).to(Mxx)
    D = torch.cat([torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.t(), Myy), 1)], 0)
    if sqrt:
        D = D.abs().sqrt()
    D = D + torch.diag(torch.full((nx + ny,), float("inf")).to(Mxx))           # exclude the cloud itself
    nearest = D.topk(k, dim=0, largest=False).indices                           # (k, nx + ny)
    votes = is_x[nearest].sum(0)
    pred = (votes >= k / 2.0).float()
    if one_way:
        pred = pred[:nx]
        is_x = pred[:nx]   # as in the reference (:229-231): the one-way variant scores the x block against itself
    tp, fp = (pred * is_x).sum(), (pred * (1 - is_x)).sum()
    fn, tn = ((1 - pred) * is_x).sum(), ((1 - pred) * (1 - is_x)).sum()
    return {"tp": tp, "fp": fp, "fn": fn, "tn": tn, "precision": tp / (tp + fp + 1e-10), "recall": tp / (tp + fn + 1e-10),
            "acc_t": tp / (tp + fn + 1e-10), "acc_f": tn / (tn + fp + 1e-10), "acc": (is_x == pred).float().mean()}


def lgan_mmd_cov(all_dist, thresh=1000):
    """all_dist (N_sample, N_ref).  The reference's variant (:251-278): lgan_mmd = mean over references of the distance to
    their closest sample; lgan_cov = number of distinct closest samples over the references (references whose closest
    sample is farther than ``thresh`` are folded onto the best-matched reference's sample) / N_ref; lgan_mmd_smp = mean
    over samples of the distance to their closest reference."""
    N_ref = all_dist.size(1)
    min_val_fromsmp, _ = torch.min(all_dist, dim=1)
    min_val, idx = torch.min(all_dist, dim=0)
    min_val, order = torch.sort(min_val)
    sorted_idx = idx[order]
    outlier = min_val > thresh
    if torch.any(outlier):
        sorted_idx[outlier] = sorted_idx[0]
    cov = torch.tensor(float(sorted_idx.unique().numel()) / float(N_ref)).to(all_dist)
    return {"lgan_mmd": min_val.mean(), "lgan_cov": cov, "lgan_mmd_smp": min_val_fromsmp.mean()}


def compute_all_metrics(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, one_way=False, mask=None):
    """MMD / COV under CD and EMD + 1-NNA accuracies (evaluation_utils.py:500-560)."""
    results = {}
    M_rs_cd, M_rs_emd = _pairwise_EMD_CD_(ref_pcs, sample_pcs, batch_size, mask_ref=mask)
    results.update({f"{k}-CD": v for k, v in lgan_mmd_cov(M_rs_cd.t()).items()})
    results.update({f"{k}-EMD": v for k, v in lgan_mmd_cov(M_rs_emd.t()).items()})
    M_rr_cd, M_rr_emd = _pairwise_EMD_CD_(ref_pcs, ref_pcs, batch_size)
    if not one_way:
        M_ss_cd, M_ss_emd = _pairwise_EMD_CD_(sample_pcs, sample_pcs, batch_size, mask_ref=mask, mask_sample=mask)
    else:
        inf = float("inf")
        M_ss_cd = torch.full((sample_pcs.shape[0],) * 2, inf).to(M_rr_cd)
        M_ss_emd = torch.full((sample_pcs.shape[0],) * 2, inf).to(M_rr_cd)
    for name, (rr, rs, ss) in (("CD", (M_rr_cd, M_rs_cd, M_ss_cd)), ("EMD", (M_rr_emd, M_rs_emd, M_ss_emd))):
        res = knn(rr, rs, ss, 1, sqrt=False, one_way=one_way)
        results.update({f"1-NN-{name}-{k}": v for k, v in res.items() if "acc" in k})
    return results

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


def EMD_CD(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, reduced=True):
    """Paired CD / EMD of sample i vs reference i."""
    assert sample_pcs.shape[0] == ref_pcs.shape[0]
    cd, emd = [], []
    for s in range(0, sample_pcs.shape[0], batch_size):
        a, b = sample_pcs[s:s + batch_size].contiguous(), ref_pcs[s:s + batch_size].contiguous()
        dl, dr = distChamferCUDA(a, b)
        cd.append(dl.mean(dim=1) + dr.mean(dim=1))
        emd.append(emd_approx(a, b))
    cd, emd = torch.cat(cd), torch.cat(emd)
    return {"MMD-CD": cd.mean() if reduced else cd, "MMD-EMD": emd.mean() if reduced else emd}


def _pairwise_EMD_CD_(sample_pcs, ref_pcs, batch_size, accelerated_cd=True, verbose=False, mask_sample=None, mask_ref=None):
    """All-pairs (N_sample, N_ref) CD and EMD matrices; optional per-point masks weight the two Chamfer directions."""
    all_cd, all_emd = [], []
    for i in range(sample_pcs.shape[0]):
        cd_row, emd_row = [], []
        for s in range(0, ref_pcs.shape[0], batch_size):
            ref = ref_pcs[s:s + batch_size].contiguous()
            smp = sample_pcs[i].view(1, -1, ref.size(2)).expand(ref.size(0), -1, -1).contiguous()
            dl, dr = distChamferCUDA(smp, ref)
            dl_mean = dl.mean(1) if mask_sample is None else (dl * mask_sample[i].unsqueeze(0)).sum(1) / mask_sample[i].sum()
            dr_mean = dr.mean(1) if mask_ref is None else (dr * mask_ref[s:s + batch_size]).sum(1) / mask_ref[s:s + batch_size].sum(1)
            cd_row.append((dl_mean + dr_mean).view(1, -1))
            emd_row.append(emd_approx(smp, ref).view(1, -1))
        all_cd.append(torch.cat(cd_row, dim=1))
        all_emd.append(torch.cat(emd_row, dim=1))
    return torch.cat(all_cd, dim=0), torch.cat(all_emd, dim=0)


def knn(Mxx, Mxy, Myy, k, sqrt=False, one_way=False):
    """Leave-one-out k-NN two-sample test on the stacked distance matrix (1-NNA for k = 1)."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat([torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.transpose(0, 1), Myy), 1)], 0)
    if sqrt:
        M = M.abs().sqrt()
    _, idx = (M + torch.diag(float("inf") * torch.ones(n0 + n1).to(Mxx))).topk(k, 0, False)
    count = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = torch.ge(count, (float(k) / 2) * torch.ones(n0 + n1).to(Mxx)).float()
    if one_way:
        pred = pred[:n0]
        label = pred[:n0]
    s = {"tp": (pred * label).sum(), "fp": (pred * (1 - label)).sum(), "fn": ((1 - pred) * label).sum(),
         "tn": ((1 - pred) * (1 - label)).sum()}
    s.update({"precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10), "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
              "acc_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10), "acc_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
              "acc": torch.eq(label, pred).float().mean()})
    return s


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

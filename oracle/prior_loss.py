"""Prior loss of the part encoder with gradients (ORACLE — test infrastructure only).

PyTorch-CPU restatement of PartEncoder.get_prior_loss (part_encoders.py:1143-1182, use_flow=True): per part, the valid shapes'
latents go FORWARD through that part's 14 coupling layers (flow.py:9-49: y1 = x1 sigmoid(s + 2) + t, logpx - log det),
log p(w) under N(0, prior_var) — with the reference's own normalisation constant, -0.5 log(2 pi) * dim PER ELEMENT
(misc.py:301-317 called with dim = 256 and summed over the 256 elements) — minus the Gaussian entropy of the posterior
(misc.py:292-295), averaged over the valid parts of a shape and over the batch, times kl_weight.  Pinned to the reference by
tests/golden/prior_loss_*.npz.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def coupling_forward(x, W, prefix, swap):
    d = x.shape[1] - x.shape[1] // 2
    if swap:
        x = torch.cat([x[:, d:], x[:, :d]], 1)
    h = F.relu(F.linear(x[:, :d], W[prefix + "0.weight"], W[prefix + "0.bias"]))
    h = F.relu(F.linear(h, W[prefix + "2.weight"], W[prefix + "2.bias"]))
    s_t = F.linear(h, W[prefix + "4.weight"], W[prefix + "4.bias"])
    out = x.shape[1] - d
    scale = torch.sigmoid(s_t[:, :out] + 2.0)
    y1 = x[:, d:] * scale + s_t[:, out:]
    logdet = torch.log(scale).sum(1)
    y = torch.cat([y1, x[:, :d]], 1) if swap else torch.cat([x[:, :d], y1], 1)
    return y, logdet


def prior_loss(W, part_code, logvar, valid, depth=14, prior_var=1.0, kl_weight=5e-4):
    """part_code (B, C, M), logvar (B, M, C), valid (B, M) torch tensors; W: dict 'flow.{i}.chain.{l}.net_s_t.*' -> tensors.
    Returns (loss, log_p_part (B, M), entropy (B, M))."""
    B, C, M = part_code.shape
    entropy = 0.5 * logvar.reshape(B * M, -1).sum(1) + 0.5 * C * (1.0 + math.log(2 * math.pi))
    log_p = torch.zeros(B, M)
    cols = []
    for i in range(M):
        x = part_code[:, :, i]
        delta = torch.zeros(B)
        for l in range(depth):
            x, ld = coupling_forward(x, W, f"flow.{i}.chain.{l}.net_s_t.", swap=(l % 2 == 0))
            delta = delta - ld
        log_pw = (-math.log(prior_var) - 0.5 * math.log(2 * math.pi) * C - x.pow(2) / (2.0 * prior_var)).sum(1)
        cols.append(torch.where(valid[:, i] == 1, log_pw - delta, torch.zeros(B)))
    log_p = torch.stack(cols, 1)
    entropy = entropy.view(B, M)
    loss_prior = ((-log_p - entropy) * valid).sum(1) / valid.sum(1)
    return kl_weight * loss_prior.mean(), log_p, entropy


def loss_and_grads(W, part_code, logvar, valid, **kw):
    """numpy in / out: loss, log_p_part, entropy, d loss / d (part_code, logvar, every flow parameter)."""
    Wt = {k: torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True) for k, a in W.items() if k.startswith("flow.")}
    z = torch.from_numpy(part_code).clone().requires_grad_(True)
    lv = torch.from_numpy(logvar).clone().requires_grad_(True)
    loss, log_p, ent = prior_loss(Wt, z, lv, torch.from_numpy(valid), **kw)
    loss.backward()
    return dict(loss=float(loss.detach()), log_p=log_p.detach().numpy(), entropy=ent.detach().numpy(), d_part_code=z.grad.numpy(),
                d_logvar=lv.grad.numpy(), grads={k: t.grad.numpy() for k, t in Wt.items()})

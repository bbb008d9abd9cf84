"""PointNetV2 part encoder in TRAIN mode (batch-statistics BatchNorm) with gradients (ORACLE — test infrastructure only).

PyTorch-CPU restatement of python/difffacto/models/encoders/pointnet.py:187-213 (per_part_mlp=True) using
``F.batch_norm(training=True)`` exactly where the reference's ``nn.BatchNorm1d`` modules sit, differentiated by torch
autograd.  Pinned to the reference class's own autograd by tests/golden/pointnet_v2_train_*.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F


def forward(Wt, x, attn, num_anchors=4, reweight_by_anchor=True, eps=1e-5, momentum=0.1, running=None):
    """Wt: dict name -> torch tensor (state_dict names).  x (B,N,3), attn (B,N,A) torch tensors.  `running`: optional dict that
    receives the updated running statistics (as nn.BatchNorm1d would leave them).  Returns m, v (B, A, zdim)."""
    B = x.shape[0]

    def bn(h, p):
        rm = Wt[p + "running_mean"].detach().clone()
        rv = Wt[p + "running_var"].detach().clone()
        y = F.batch_norm(h, rm, rv, Wt[p + "weight"], Wt[p + "bias"], training=True, momentum=momentum, eps=eps)
        if running is not None:
            running[p + "running_mean"], running[p + "running_var"] = rm, rv
        return y

    h = x.transpose(1, 2)
    for i in (1, 2, 3, 4):
        h = bn(F.conv1d(h, Wt[f"conv{i}.weight"], Wt[f"conv{i}.bias"]), f"bn{i}.")
        if i < 4:
            h = F.relu(h)
    w = h.unsqueeze(-1) * attn.unsqueeze(1)
    if reweight_by_anchor:
        w = w * num_anchors
    pooled = torch.max(w, 2, keepdim=True)[0].view(B, 512, num_anchors)
    z = pooled.transpose(1, 2).reshape(B, -1, 1)
    outs = []
    for name in ("mlp_m", "mlp_v"):
        t = F.relu(bn(F.conv1d(z, Wt[name + ".0.weight"], Wt[name + ".0.bias"], groups=num_anchors), name + ".1."))
        t = F.relu(bn(F.conv1d(t, Wt[name + ".3.weight"], Wt[name + ".3.bias"], groups=num_anchors), name + ".4."))
        t = F.conv1d(t, Wt[name + ".6.weight"], Wt[name + ".6.bias"], groups=num_anchors)
        outs.append(t.reshape(B, num_anchors, -1))
    return outs[0], outs[1]


def outputs_and_grads(W, x, attn, dm, dv, **kw):
    """numpy in / out: m, v, updated running statistics, and d (sum(m dm) + sum(v dv)) / d (every trainable parameter)."""
    Wt = {k: torch.from_numpy(np.ascontiguousarray(a)).clone() for k, a in W.items()}
    for k, t in Wt.items():
        if "running" not in k:
            t.requires_grad_(True)
    running = {}
    m, v = forward(Wt, torch.from_numpy(x), torch.from_numpy(attn), running=running, **kw)
    ((m * torch.from_numpy(dm)).sum() + (v * torch.from_numpy(dv)).sum()).backward()
    return dict(m=m.detach().numpy(), v=v.detach().numpy(), running={k: t.numpy() for k, t in running.items()},
                grads={k: t.grad.numpy() for k, t in Wt.items() if t.requires_grad})

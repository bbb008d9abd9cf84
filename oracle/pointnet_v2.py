"""numpy fp32 restatement of the ``PointNetV2`` masked max-pool part encoder (ORACLE — test only).

SURVEY.md §8 A17: python/difffacto/models/encoders/pointnet.py:124-213 with ``per_part_mlp=True`` (configs/gen_chair.py:8-13),
BatchNorm in eval mode (running statistics, eps 1e-5):

    x (B,N,3) -> conv1..4 (3 -> 128 -> 128 -> 256 -> 512, BN after each, ReLU after the first three)      :189-193
    weighted = x[..., None] * attn_weight[:, None] (* num_anchors if reweight_by_anchor); max over N      :194-198
    per part: grouped Conv1d stack 512 -> 256 -> 128 -> zdim (BN + ReLU after the first two), twice (m, v) :200-203

``W`` maps the module's ``state_dict`` names (``conv1.weight``, ``bn1.running_mean``, ``mlp_m.0.weight`` ...) to numpy arrays.
Pinned by tests/golden/pointnet_v2_*.npz (the reference class run on CPU, tests/golden/make_golden.py).
"""
import numpy as np

F32 = np.float32
EPS = 1e-5


def _bn(x, W, p):
    inv = (F32(1) / np.sqrt(W[p + "running_var"] + F32(EPS))).astype(F32)
    sh = [1, -1] + [1] * (x.ndim - 2)
    return ((x - W[p + "running_mean"].reshape(sh)) * inv.reshape(sh) * W[p + "weight"].reshape(sh) + W[p + "bias"].reshape(sh)).astype(F32)


def _conv1(x, w, b):
    """x (B, Cin, L), w (Cout, Cin, 1)."""
    return (np.einsum("oc,bcl->bol", w[:, :, 0], x, optimize=True) + b[None, :, None]).astype(F32)


def _gconv1(x, w, b, groups):
    """grouped 1x1 conv: x (B, G*Cin, L), w (G*Cout, Cin, 1)."""
    B, C, L = x.shape
    cin, cout = C // groups, w.shape[0] // groups
    xg = x.reshape(B, groups, cin, L)
    wg = w[:, :, 0].reshape(groups, cout, cin)
    return (np.einsum("goc,bgcl->bgol", wg, xg, optimize=True).reshape(B, groups * cout, L) + b[None, :, None]).astype(F32)


def forward(W, x, attn_weight, num_anchors=4, reweight_by_anchor=True):
    """x (B,N,3), attn_weight (B,N,num_anchors) -> m, v (B, num_anchors, zdim)."""
    B = x.shape[0]
    h = np.ascontiguousarray(x.transpose(0, 2, 1)).astype(F32)
    for i in (1, 2, 3, 4):
        h = _bn(_conv1(h, W[f"conv{i}.weight"], W[f"conv{i}.bias"]), W, f"bn{i}.")
        if i < 4:
            h = np.maximum(h, 0)
    wx = (h[:, :, :, None] * attn_weight[:, None, :, :]).astype(F32)
    if reweight_by_anchor:
        wx = (wx * F32(num_anchors)).astype(F32)
    pooled = wx.max(axis=2)                                   # (B, 512, A)
    z = np.ascontiguousarray(pooled.transpose(0, 2, 1)).reshape(B, -1, 1)
    outs = []
    for name in ("mlp_m", "mlp_v"):
        t = z
        t = np.maximum(_bn(_gconv1(t, W[name + ".0.weight"], W[name + ".0.bias"], num_anchors), W, name + ".1."), 0)
        t = np.maximum(_bn(_gconv1(t, W[name + ".3.weight"], W[name + ".3.bias"], num_anchors), W, name + ".4."), 0)
        t = _gconv1(t, W[name + ".6.weight"], W[name + ".6.bias"], num_anchors)
        outs.append(t.reshape(B, num_anchors, -1))
    return outs[0], outs[1]

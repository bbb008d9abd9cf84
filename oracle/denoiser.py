"""numpy fp32 restatement of the cross-diffusion denoiser ``TransformerNet`` (ORACLE — test only).

Follows the reference op by op, unfused, in fp32:

* ``timestep_embedding``      python/difffacto/models/diffusions/nets/utils.py:7-24
* ``GEGLU`` / ``FeedForward`` python/difffacto/models/diffusions/nets/attention.py:50-57,77-94
* ``CrossAttention.forward``  attention.py:179-204 (mask fill ``-finfo.max`` :192-197)
* ``BasicTransformerBlock._forward`` attention.py:296-306 (``single_attn=True``, ``adaln=False``)
* ``TransformerNet.forward`` / ``_forward_attn`` attention.py:385-409 / 411-440 with the shipped
  config (configs/gen_chair.py:54-70): class_cond, use_linear, cat_params_to_x, single_attn,
  cat_class_to_x, mask_out_unreferenced_code; not add_class_cond / add_t_to_x / context_proj /
  include_std / res.

``W`` is a dict of fp32 numpy arrays keyed by the reference ``state_dict`` names relative to
``diffusion.model.`` (e.g. ``transformer_blocks.0.attn2.to_q.weight``).
"""
import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32
LN_EPS = 1e-5  # torch.nn.LayerNorm default


def timestep_embedding(t, dim=256, max_period=10000):
    """utils.py:7-24 — ``[cos(t*f_k) | sin(t*f_k)]``, f_k = exp(-ln(max_period)*k/half)."""
    t = np.asarray(t)
    half = dim // 2
    freqs = np.exp((-math.log(max_period) * np.arange(half, dtype=F32) / F32(half)).astype(F32)).astype(F32)
    # reference: timesteps[:, None].to(timesteps.dtype) * freqs[None]  (int64 * fp32 -> fp32)
    args = (t.astype(F32)[:, None] * freqs[None]).astype(F32)
    return np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32)


def layer_norm(x, w, b):
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(LN_EPS)) * w + b).astype(F32)


def gelu(x):
    """F.gelu default = exact erf form (attention.py:57)."""
    return (x * F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))).astype(F32)


def feed_forward_glu(x, W, prefix):
    """FeedForward(glu=True): GEGLU proj -> (Dropout eval) -> Linear  (attention.py:77-94)."""
    h = linear(x, W[prefix + "net.0.proj.weight"], W[prefix + "net.0.proj.bias"])
    a, g = np.split(h, 2, axis=-1)
    h = (a * gelu(g)).astype(F32)
    return linear(h, W[prefix + "net.2.weight"], W[prefix + "net.2.bias"])


def cross_attention(x, context, mask, W, prefix, heads=8):
    """attention.py:179-204.  x (B,N,C)  context (B,J,Cc)  mask (B,J) or None."""
    B, N, _ = x.shape
    J = context.shape[1]
    q = linear(x, W[prefix + "to_q.weight"])
    k = linear(context, W[prefix + "to_k.weight"])
    v = linear(context, W[prefix + "to_v.weight"])
    inner = q.shape[-1]
    d = inner // heads
    scale = F32(d ** -0.5)
    q = q.reshape(B, N, heads, d).transpose(0, 2, 1, 3)  # B h N d
    k = k.reshape(B, J, heads, d).transpose(0, 2, 1, 3)
    v = v.reshape(B, J, heads, d).transpose(0, 2, 1, 3)
    sim = (np.einsum("bhid,bhjd->bhij", q, k).astype(F32) * scale).astype(F32)
    if mask is not None:
        assert mask.shape == (B, J)
        keep = mask.astype(bool)[:, None, None, :]
        sim = np.where(keep, sim, F32(-np.finfo(np.float32).max)).astype(F32)
    sim = sim - sim.max(axis=-1, keepdims=True)
    e = np.exp(sim).astype(F32)
    p = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    out = np.einsum("bhij,bhjd->bhid", p, v).astype(F32)
    out = out.transpose(0, 2, 1, 3).reshape(B, N, inner)
    return linear(out, W[prefix + "to_out.0.weight"], W[prefix + "to_out.0.bias"])


def transformer_block(x, context, mask, W, prefix):
    """attention.py:296-306 with single_attn=True, adaln=False."""
    x = cross_attention(layer_norm(x, W[prefix + "norm2.weight"], W[prefix + "norm2.bias"]),
                        context, mask, W, prefix + "attn2.") + x
    x = feed_forward_glu(layer_norm(x, W[prefix + "norm3.weight"], W[prefix + "norm3.bias"]),
                         W, prefix + "ff.") + x
    return x.astype(F32)


def time_embed(W, t):
    """``self.time_embed(timestep_embedding(t, 256))`` attention.py:393 / :357."""
    return feed_forward_glu(timestep_embedding(t, 256), W, "time_embed.")


def build_context(W, t, ctx_list, n_class=4):
    """attention.py:386-397: ctx (B,J,522) = [part_code|mean|var|eye(J)|t_embed]."""
    ctx = np.concatenate(ctx_list, axis=1)            # (B, 262, J)
    ctx = np.ascontiguousarray(ctx.transpose(0, 2, 1))  # (B, J, 262)
    B = ctx.shape[0]
    eye = np.broadcast_to(np.eye(n_class, dtype=F32)[None], (B, n_class, n_class))
    ctx = np.concatenate([ctx, eye], axis=-1)
    t_emb = time_embed(W, t)                          # (B,256)
    ctx = np.concatenate([ctx, np.broadcast_to(t_emb[:, None, :], (B, ctx.shape[1], 256))], axis=-1)
    return ctx.astype(F32)


def depth_of(W):
    d = 0
    while f"transformer_blocks.{d}.norm2.weight" in W:
        d += 1
    return d


def transformer_net_forward(W, x, t, ctx_list, anchors, variances, valid_id, anchor_assignment, n_class=4):
    """TransformerNet.forward (attention.py:385-409).

    x (B,3,N) fp32; t (B,) int; ctx_list [(B,256,J),(B,6,J)]; anchors, variances (B,N,3)
    (already transposed as the caller does at anchored_diffusion.py:261); valid_id (B,J);
    anchor_assignment (B,N) int32.  Returns eps (B,3,N).
    """
    x = np.asarray(x, dtype=F32)
    ctx = build_context(W, t, ctx_list, n_class)
    assert ctx.shape[-1] == 522
    xin = np.concatenate([x, anchors.transpose(0, 2, 1), variances.transpose(0, 2, 1)], axis=1)
    onehot = np.eye(n_class, dtype=F32)[np.asarray(anchor_assignment).astype(np.int64)]  # (B,N,J)
    xin = np.concatenate([xin, onehot.transpose(0, 2, 1)], axis=1).astype(F32)       # (B,13,N)
    assert xin.shape[1] == 13
    # _forward_attn (attention.py:411-440), use_linear=True
    h = np.ascontiguousarray(xin.transpose(0, 2, 1))                                   # (B,N,13)
    h = linear(h, W["proj_in.weight"], W["proj_in.bias"])
    h = layer_norm(h, W["pre_norm.weight"], W["pre_norm.bias"])
    for i in range(depth_of(W)):
        h = transformer_block(h, ctx, valid_id, W, f"transformer_blocks.{i}.")
    h = layer_norm(h, W["post_norm.weight"], W["post_norm.bias"])
    h = linear(h, W["proj_out.weight"], W["proj_out.bias"])
    return np.ascontiguousarray(h.transpose(0, 2, 1)).astype(F32)                      # (B,3,N)

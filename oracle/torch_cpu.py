"""PyTorch-CPU restatement of the denoiser evaluation and the anchored DDPM step (ORACLE — test / baseline only).

Same unfused op sequence as the reference's modules, written with the torch CPU ops the reference itself would run
(`F.linear`, `F.layer_norm`, `F.gelu`, `softmax`, `einsum`): TransformerNet.forward attention.py:385-440, CrossAttention
:179-204, GEGLU FeedForward :50-94, timestep_embedding nets/utils.py:7-24, p_mean_variance / p_sample
anchored_diffusion.py:227-395, 450-484.  It exists for `bench.py`'s `cpu_baseline` leg — "the reference's CPU PyTorch path
timed on the box's host cores" cannot be the reference itself (it does not travel to the GPU box), so this restatement,
pinned to the same reference goldens as the numpy oracle (tests/test_oracle_golden.py), stands in for it — and as a second,
independent oracle.  Never imported by the product path.
"""
import math

import torch
import torch.nn.functional as F


def _w(W, k):
    return W[k]


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def feed_forward(x, W, p, drop=None):
    """drop: explicit dropout factors (0 or 1/(1-p)) behind the GEGLU (attention.py:84), or None (eval mode)."""
    a, g = F.linear(x, W[p + "net.0.proj.weight"], W[p + "net.0.proj.bias"]).chunk(2, dim=-1)
    h = a * F.gelu(g)
    if drop is not None:
        h = h * drop
    return F.linear(h, W[p + "net.2.weight"], W[p + "net.2.bias"])


def cross_attention(x, context, mask, W, p, heads=8, drop=None):
    B, N, _ = x.shape
    J = context.shape[1]
    q, k, v = F.linear(x, W[p + "to_q.weight"]), F.linear(context, W[p + "to_k.weight"]), F.linear(context, W[p + "to_v.weight"])
    d = q.shape[-1] // heads
    q, k, v = (t.reshape(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    if mask is not None:
        sim = sim.masked_fill(~mask.bool()[:, None, None, :], -torch.finfo(sim.dtype).max)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v).transpose(1, 2).reshape(B, N, heads * d)
    out = F.linear(out, W[p + "to_out.0.weight"], W[p + "to_out.0.bias"])
    return out if drop is None else out * drop   # to_out = Sequential(Linear, Dropout)  (attention.py:177)


def transformer_net_forward(W, x, t, ctx_list, anchors, variances, valid_id, anchor_assignment, n_class=4, depth=5, drops=None):
    """Arguments as oracle.denoiser.transformer_net_forward, torch CPU tensors; returns eps (B,3,N)."""
    B = x.shape[0]
    ctx = torch.cat(ctx_list, dim=1).transpose(1, 2)
    ctx = torch.cat([ctx, torch.eye(n_class)[None].expand(B, -1, -1)], dim=-1)
    drops = drops or {}   # explicit dropout factors: {"te": (B,1024), ("attn", i): (B,N,128), ("ff", i): (B,N,512)}
    t_emb = feed_forward(timestep_embedding(t), W, "time_embed.", drops.get("te"))
    ctx = torch.cat([ctx, t_emb[:, None, :].expand(-1, ctx.shape[1], -1)], dim=-1)
    onehot = F.one_hot(anchor_assignment.long(), n_class).float()
    h = torch.cat([x.transpose(1, 2), anchors, variances, onehot], dim=-1)
    h = F.layer_norm(F.linear(h, W["proj_in.weight"], W["proj_in.bias"]), (128,), W["pre_norm.weight"], W["pre_norm.bias"])
    for i in range(depth):
        p = f"transformer_blocks.{i}."
        h = cross_attention(F.layer_norm(h, (128,), W[p + "norm2.weight"], W[p + "norm2.bias"]), ctx, valid_id, W, p + "attn2.",
                            drop=drops.get(("attn", i))) + h
        h = feed_forward(F.layer_norm(h, (128,), W[p + "norm3.weight"], W[p + "norm3.bias"]), W, p + "ff.", drops.get(("ff", i))) + h
    h = F.layer_norm(h, (128,), W["post_norm.weight"], W["post_norm.bias"])
    return F.linear(h, W["proj_out.weight"], W["proj_out.bias"]).transpose(1, 2)


def p_sample(tables, W, x, t, anchors, ctx_list, variance, anchor_assignment, valid_id, noise):
    """One reverse step; ``tables`` = oracle.diffusion.Tables (float64 numpy, cast to fp32 at use like the reference)."""
    B = x.shape[0]
    f = lambda name: float(getattr(tables, name).astype("float32")[t])
    eps = transformer_net_forward(W, x, torch.full((B,), t, dtype=torch.long), ctx_list, anchors.transpose(1, 2),
                                  variance.transpose(1, 2), valid_id, anchor_assignment)
    L = torch.sqrt(variance)
    x0 = f("sqrt_recip_alphas_cumprod") * (x - anchors) + anchors - f("sqrt_recipm1_alphas_cumprod") * L * eps
    mean = f("posterior_mean_coef1") * x0 + f("posterior_mean_coef2") * x + f("posterior_mean_coef3") * anchors
    nz = 1.0 if t != 0 else 0.0
    return mean + nz * torch.sqrt(f("posterior_variance") * variance) * noise, eps

"""numpy fp32 restatement of the per-batch latent sampler (ORACLE — test only).

SURVEY.md §8 row F2: ``PartEncoder.sample_latents`` for the shipped gen configs
(configs/gen_chair.py:6-47).  Follows the reference op by op, unfused, in fp32:

* ``CouplingLayer.forward(reverse=True)``   python/difffacto/models/encoders/flow.py:21-47
* ``SequentialFlow.forward(reverse=True)``  flow.py:58-72 (layers visited last -> first);
  ``build_latent_flow`` flow.py:75-79 (``swap = (i % 2 == 0)``)
* ``PartAlignerTransformer.forward`` / ``_forward_attn``  part_encoders.py:88-143 with
  class_cond + add_class_cond (class embedding added after proj_in), cimle with
  cond_noise_type 0 (noise * noise_scale concatenated to every token), use_linear,
  single_attn (self-attention over the n_class tokens, keys masked by valid_id); pre_norm is
  NOT applied on this configuration (part_encoders.py:115-131)
* ``PartEncoder.sample_latents``            part_encoders.py:1052-1110 (fixed_id mixing, K-fold
  repeat, seg-mask ids) with ``PartEncoderForTransformerDecoder.prepare_ctx`` :1317-1326 and
  ``gather_all`` :417-428

``W`` is a dict of fp32 numpy arrays keyed by the reference ``state_dict`` names relative to
``encoder.`` (e.g. ``flow.0.chain.3.net_s_t.2.weight``, ``part_aligner.proj_in.weight``).
The random draws are inputs (``w_noise``, ``aligner_noise``) so that tests can replay the
reference's own torch.randn sequence.
"""
import numpy as np

from .denoiser import F32, cross_attention, feed_forward_glu, layer_norm, linear


def coupling_reverse(x, W, prefix, swap):
    """flow.py:21-47 with reverse=True: y1 = (x2 - shift) / sigmoid(s + 2)."""
    D = x.shape[1]
    d = D - D // 2
    if swap:
        x = np.concatenate([x[:, d:], x[:, :d]], axis=1)
    h = np.maximum(linear(x[:, :d], W[prefix + "net_s_t.0.weight"], W[prefix + "net_s_t.0.bias"]), 0)
    h = np.maximum(linear(h, W[prefix + "net_s_t.2.weight"], W[prefix + "net_s_t.2.bias"]), 0)
    s_t = linear(h, W[prefix + "net_s_t.4.weight"], W[prefix + "net_s_t.4.bias"])
    out_dim = D - d
    scale = (F32(1) / (F32(1) + np.exp(-(s_t[:, :out_dim] + F32(2.0))))).astype(F32)
    shift = s_t[:, out_dim:]
    y1 = ((x[:, d:] - shift) / scale).astype(F32)
    if not swap:
        return np.concatenate([x[:, :d], y1], axis=1)
    return np.concatenate([y1, x[:, :d]], axis=1)


def flow_reverse(x, W, part, depth):
    """SequentialFlow(reverse=True) of part ``part`` (flow.py:58-72)."""
    for i in range(depth - 1, -1, -1):
        x = coupling_reverse(x, W, f"flow.{part}.chain.{i}.", swap=(i % 2 == 0))
    return x


def flow_depth(W, part=0):
    d = 0
    while f"flow.{part}.chain.{d}.net_s_t.0.weight" in W:
        d += 1
    return d


def aligner_depth(W):
    d = 0
    while f"part_aligner.transformer_blocks.{d}.norm2.weight" in W:
        d += 1
    return d


def part_aligner_forward(W, part_code, valid_id, noise, noise_scale=100.0, heads=8):
    """part_encoders.py:88-143.  part_code (B,zdim,J)  valid_id (B,J)  noise (B,noise_dim)
    -> mean (B,3,J), logvar (B,3,J)."""
    P = "part_aligner."
    B, _, J = part_code.shape
    nz = (noise * F32(noise_scale)).astype(F32)
    x = np.concatenate([part_code, np.repeat(nz[:, :, None], J, axis=2)], axis=1)     # (B, zdim+noise_dim, J)
    x = np.ascontiguousarray(x.transpose(0, 2, 1))                                    # b c n -> b n c
    x = linear(x, W[P + "proj_in.weight"], W[P + "proj_in.bias"])
    x = (x + W[P + "class_emb.weight"][None]).astype(F32)
    # part_encoders.py:115-131: with cimle and cond_noise_type 0 NO branch applies pre_norm (the `else`
    # that holds it pairs with `if self.cimle`); its parameters exist in the state_dict but are unused.
    for i in range(aligner_depth(W)):
        p = f"{P}transformer_blocks.{i}."
        xn = layer_norm(x, W[p + "norm2.weight"], W[p + "norm2.bias"])
        x = (cross_attention(xn, xn, valid_id, W, p + "attn2.", heads=heads) + x).astype(F32)
        x = (feed_forward_glu(layer_norm(x, W[p + "norm3.weight"], W[p + "norm3.bias"]), W, p + "ff.") + x).astype(F32)
    x = layer_norm(x, W[P + "post_norm.weight"], W[P + "post_norm.bias"])
    x = linear(x, W[P + "proj_out.weight"], W[P + "proj_out.bias"])                   # (B, J, 6)
    h = np.ascontiguousarray(x.transpose(0, 2, 1))                                    # b n c -> b c n
    return h[:, :3].copy(), h[:, 3:].copy()


def seg_mask_ids(valid_id, sample_points):
    """part_encoders.py:1105-1106."""
    valid_id = np.asarray(valid_id, dtype=F32)
    B, J = valid_id.shape
    ids = np.arange(J, dtype=F32)[None] * valid_id + np.argmax(valid_id, axis=1)[:, None].astype(F32) * (1 - valid_id)
    seg = np.repeat(ids.astype(np.int32)[:, :, None], sample_points // J, axis=2).reshape(B, -1)
    return np.ascontiguousarray(seg)


def sample_latents(W, w_noise, aligner_noise, valid_id, fixed_id, K, sample_points, prior_var=1.0,
                   noise_scale=100.0, log_scale_var=0.0, part_code=None):
    """part_encoders.py:1052-1110 (use_flow, cimle, no selective sampling).

    w_noise (S,zdim,J) standard normal; aligner_noise (S*K,noise_dim); valid_id (S,J); fixed_id (J,).
    Returns the reference's 6-tuple as a dict."""
    S, _, J = w_noise.shape
    if part_code is None:
        part_code = (w_noise * F32(np.sqrt(prior_var))).astype(F32)
        depth = flow_depth(W)
        if depth:
            part_code = np.stack([flow_reverse(np.ascontiguousarray(part_code[..., i]), W, i, depth)
                                  for i in range(J)], axis=-1)
    fixed_id = np.asarray(fixed_id, dtype=F32)
    valid_id = np.asarray(valid_id, dtype=F32)
    noise = np.asarray(aligner_noise, dtype=F32)
    fixed_codes = part_code[0][None]
    fixed_valid = np.clip(valid_id[0][None] + fixed_id[None], 0, 1)
    part_code = (part_code * (1 - fixed_id)[None, None] + fixed_id[None, None] * fixed_codes).astype(F32)
    valid_id = (valid_id * (1 - fixed_id)[None] + fixed_id[None] * fixed_valid).astype(F32)
    if np.any(fixed_id == 1):
        noise = np.broadcast_to(noise.reshape(S, K, -1)[0][None], (S, K, noise.shape[-1])).reshape(S * K, -1)
    part_code = np.repeat(part_code, K, axis=0)
    valid_id = np.repeat(valid_id, K, axis=0)
    mean, logvar = part_aligner_forward(W, part_code, valid_id, noise, noise_scale=noise_scale)
    seg = seg_mask_ids(valid_id, sample_points)
    lv = (logvar + F32(log_scale_var)).astype(F32)
    idx = seg[:, None, :].astype(np.int64)
    mean_pp = np.take_along_axis(mean, np.broadcast_to(idx, (mean.shape[0], 3, seg.shape[1])), axis=2)
    logvar_pp = np.take_along_axis(lv, np.broadcast_to(idx, (mean.shape[0], 3, seg.shape[1])), axis=2)
    ctx = [part_code, np.concatenate([mean, np.exp(lv)], axis=1).astype(F32)]
    return {"ctx": ctx, "mean_per_point": mean_pp, "logvar_per_point": logvar_pp, "seg_mask": seg,
            "valid_id": valid_id, "part_code": part_code, "mean": mean, "logvar": logvar, "noise": noise}

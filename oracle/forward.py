"""CPU restatement (numpy on top of the other oracle pieces) of the reference's top-level compositions.  TEST INFRASTRUCTURE ONLY.

* ``encoder_forward``   PartEncoder.forward with a part aligner (python/difffacto/models/encoders/part_encoders.py:1185-1260)
* ``sample_noise``      PartEncoder.sample_noise (:388-414)
* ``fit_loss``          PartEncoder.get_fit_loss, types 4 and 1 (:489-521)
* ``anchor_forward``    AnchorDiffAE.forward in eval mode (python/difffacto/models/networks/anchor_gen.py:970-1136): the gen branch
                        (:1034-1084) and the encode -> decode "sample" mode (:1085-1134), cIMLE configuration

Random draws are an explicit list consumed in the reference's call order (the ``draw_{i}`` arrays of
tests/golden/make_golden_forward.py).  Pinned by tests/test_oracle_golden.py against forward_*.npz / encoder_fwd_*.npz, which the
reference's own classes produced.
"""
import math

import numpy as np
import torch

from . import diffusion as odf
from . import latents as ol
from . import pointnet_v2 as opv
from . import prior_loss as opl

F32 = np.float32


def fit_loss(mean, logvar, valid, gt_shift, gt_var, fit_loss_type=4):
    """part_encoders.py:514-519 (type 4) / :495-500 (type 1) -> (B,)."""
    if fit_loss_type == 4:
        pred, target = np.concatenate([mean, logvar], 1), np.concatenate([gt_shift, np.log(gt_var)], 1)
    else:
        pred, target = np.concatenate([mean, np.exp(logvar)], 1), np.concatenate([gt_shift, gt_var], 1)
    d = ((pred - target) ** 2).astype(F32) * valid[:, None, :]
    return (d.sum(axis=(1, 2)) / valid.sum(axis=1)).astype(F32)


def _split(W):
    Wpn = {k[len("encoder."):]: v for k, v in W.items() if k.startswith("encoder.")}
    Wlat = {k: v for k, v in W.items() if not k.startswith("encoder.")}
    return Wpn, Wlat


def _gt(batch, origin_scale=False):
    B = batch["ref"].shape[0]
    gt_shift = batch.get("part_shift", np.zeros((B, 3, 4), F32)).astype(F32)
    gt_var = batch.get("part_scale", np.ones((B, 3, 4), F32)).astype(F32)
    return gt_shift, gt_var if origin_scale else (gt_var ** 2).astype(F32)


def _reparameterize(Wpn, x, attn, draws):
    m, lv = opv.forward(Wpn, x, attn)                                                       # (B, J, zdim) each
    eps = draws.pop(0)
    assert eps.shape == lv.shape, (eps.shape, lv.shape)
    return m, lv, np.ascontiguousarray((m + np.exp(F32(0.5) * lv) * eps).astype(F32).transpose(0, 2, 1))


def encoder_forward(W, batch, draws, noise=None, noise_scale=100.0, kl_weight=0.0, prior_var=1.0, epoch=-1):
    """W: 'encoder.*' (PointNetV2) + 'flow.*' + 'part_aligner.*' numpy weights (the encoder's state_dict names).  ``draws``: list,
    consumed from the front.  Returns a dict with the reference's six outputs."""
    Wpn, Wlat = _split(W)
    valid, seg = batch["present"].astype(F32), batch["ref_seg_mask"].astype(np.int64)
    gt_shift, gt_var = _gt(batch)
    B = valid.shape[0]
    if noise is None:
        noise = batch["noise"][:, None]
    m, lv, part_code = _reparameterize(Wpn, batch["input"], batch["ref_attn_map"], draws)
    Wt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in Wlat.items() if k.startswith("flow.")}
    loss, log_p, ent = opl.prior_loss(Wt, torch.from_numpy(part_code), torch.from_numpy(lv), torch.from_numpy(valid), depth=ol.flow_depth(Wlat),
                                      prior_var=prior_var, kl_weight=kl_weight)
    nv = valid.sum(0)
    losses = {"prior_loss": np.array(float(loss), F32), "kl_weight": np.array([kl_weight], F32)}
    for i in range(valid.shape[1]):
        losses[f"log_p_part_{i}"] = np.array(float((log_p.numpy()[:, i] * valid[:, i]).sum() / nv[i]), F32)
        losses[f"entropy_{i}"] = np.array(float((ent.numpy()[:, i] * valid[:, i]).sum() / nv[i]), F32)
        losses[f"part_{i}_mean"] = np.array(m.mean(2).sum(0)[i] / nv[i], F32)
        losses[f"part_{i}_logvar"] = np.array(lv.mean(2).sum(0)[i] / nv[i], F32)
    num = noise.shape[1]
    noise = noise.reshape(B * num, -1).astype(F32)
    rep = lambda a: np.repeat(a, num, axis=0)
    part_code, valid, seg, gt_shift, gt_var = map(rep, (part_code, valid, seg, gt_shift, gt_var))
    mean, logvar = ol.part_aligner_forward(Wlat, part_code, valid, noise, noise_scale=noise_scale)
    mean_pp, logvar_pp = odf.gather_params(seg, mean, logvar)
    flag_pp = np.take_along_axis(valid[:, None, :], seg[:, None, :], axis=2).astype(F32)
    losses["fit_loss"] = fit_loss(mean, logvar, valid, gt_shift, gt_var)
    ctx = [part_code, np.concatenate([mean, np.exp(logvar)], 1).astype(F32)]
    return dict(ctx=ctx, mean_pp=mean_pp, logvar_pp=logvar_pp, flag_pp=flag_pp, losses=losses, part_code=part_code, mean=mean, logvar=logvar,
                noise=noise)


def sample_noise(W, batch, draws, num, noise_scale=100.0):
    """part_encoders.py:388-414 -> (noise (B, num, noise_dim), id (B,))."""
    Wpn, Wlat = _split(W)
    valid = batch["present"].astype(F32)
    gt_shift, gt_var = _gt(batch)
    B = valid.shape[0]
    _, _, part_code = _reparameterize(Wpn, batch["input"], batch["attn_map"], draws)
    noise = draws.pop(0)
    assert noise.shape == (B * num, 32), noise.shape
    rep = lambda a: np.repeat(a, num, axis=0)
    part_code, valid, gt_shift, gt_var = map(rep, (part_code, valid, gt_shift, gt_var))
    mean, logvar = ol.part_aligner_forward(Wlat, part_code, valid, noise, noise_scale=noise_scale)
    fit = fit_loss(mean, logvar, valid, gt_shift, gt_var)
    return noise.reshape(B, num, -1), fit.reshape(B, num).argmin(1)


def _fold(v, h):
    return v.reshape(v.shape[0] // h, h, *v.shape[1:])


def anchor_forward(W_enc, W_den, batch, draws, T, npoints, K, gen, ret_interval=5, noise_scale=100.0, fixed_id=(0, 0, 0, 0)):
    """AnchorDiffAE.forward, eval, cIMLE, ret_traj: (dict of numpy arrays, name)."""
    draws = list(draws)
    tb = odf.Tables(T)
    B, N, C = batch["ref"].shape
    valid_in = batch["present"].astype(F32)
    enc = encoder_forward(W_enc, batch, draws, noise_scale=noise_scale)                      # anchor_gen.py:995 (every val batch)
    _, Wlat = _split(W_enc)

    def run_decode(ctx, mean_pp, var_pp, seg, valid):
        x_T = draws.pop(0)
        steps = np.stack([draws.pop(0) for _ in range(T)])
        return odf.decode(tb, W_den, mean_pp, ctx, var_pp, seg, valid, x_T, steps, ret_traj=True, ret_interval=ret_interval)

    if gen:
        w, an = draws.pop(0), draws.pop(0)
        lat = ol.sample_latents(Wlat, w, an, valid_in, list(fixed_id), K, npoints, noise_scale=noise_scale)
        mean_pp, var_pp, seg, valid = lat["mean_per_point"], np.exp(lat["logvar_per_point"]).astype(F32), lat["seg_mask"], lat["valid_id"]
        _pred = run_decode(lat["ctx"], mean_pp, var_pp, seg, valid)
        pri = draws.pop(0)
        priors = (pri * np.sqrt(var_pp.transpose(0, 2, 1)) + mean_pp.transpose(0, 2, 1)).astype(F32)
        pred = {}
        for i in range(K):
            for k, v in _pred.items():
                pred[f"{k}_sample {i}"] = _fold(v, K)[:, i]
        for i in range(K):
            pred[f"sample prior {i}"] = priors.reshape(B, K, npoints, C)[:, i]
        pred["pred"] = _fold(_pred["pred"], K)[:, 0]
        pred["pred_seg_mask"] = _fold(seg, K)[:, 0]
        pred["anchors"] = _fold(mean_pp, K)[:, 0].transpose(0, 2, 1)
        name = "gen_fixed" + "".join(str(i) for i in fixed_id)
    else:
        noise, _ = sample_noise(W_enc, batch, draws, K, noise_scale=noise_scale)
        enc = encoder_forward(W_enc, batch, draws, noise=noise, noise_scale=noise_scale)
        seg = np.repeat(batch["ref_seg_mask"].astype(np.int64), K, axis=0)
        valid = np.repeat(valid_in, K, axis=0)
        mean_pp, var_pp = enc["mean_pp"], np.exp(enc["logvar_pp"]).astype(F32)
        _pred = run_decode(enc["ctx"], mean_pp, var_pp, seg, valid)
        pred = {}
        for i in range(K):
            for k, v in _pred.items():
                pred[f"{k}_sample {i}"] = _fold(v, K)[:, i]
        for i in range(K):
            pri = draws.pop(0).transpose(0, 2, 1)                                               # randn_like(var (R,3,N)).transpose(1,2)  :1110
            priors = (pri * np.sqrt(var_pp.transpose(0, 2, 1)) + mean_pp.transpose(0, 2, 1)).astype(F32)
            pred[f"sample prior {i}"] = priors.reshape(B, K, npoints, C)[:, i]
            pred[f"noise latent {i}"] = enc["noise"].reshape(B, K, 32)[:, i]
            pred[f"sample {i} mean"] = enc["mean"].reshape(B, K, 3, 4)[:, i]
            pred[f"sample {i} logvar"] = enc["logvar"].reshape(B, K, 3, 4)[:, i]
        pred["pred"] = _fold(_pred["pred"], K)[:, 0]
        pred["pred_seg_mask"] = _fold(seg, K)[:, 0]
        pred["anchors"] = _fold(mean_pp, K)[:, 0].transpose(0, 2, 1)
        pred["part_latents"] = _fold(enc["part_code"], K)[:, 0]
        pred["valid_id"] = _fold(valid, K)[:, 0]
        pred["token"] = batch["token"]
        valid_in = valid                                                                        # :1126 writes the repeated valid_id
        name = "sample"
    pred.update({"input": batch["input"], "input_ref": batch["ref"], "ref_seg_mask": batch["ref_seg_mask"], "seg_mask": batch["seg_mask"],
                 "present": valid_in, "shift": batch["shift"], "scale": batch["scale"]})
    assert not draws, f"{len(draws)} draws left"
    return pred, name

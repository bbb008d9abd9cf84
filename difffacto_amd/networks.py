"""Host-side mirror of the reference's top-level model ``MODELS['AnchorDiffAE']`` (python/difffacto/models/networks/anchor_gen.py:29-1136)
for the configurations it ships: same constructor arguments, sub-module names (``encoder.*`` / ``diffusion.*`` state_dict keys, so
``pretrained/*.pth`` loads), methods and — the contract ``Runner.val`` writes to disk (runner/runner.py:353-377) — the same output
dicts from ``forward``:

* eval + ``gen``            -> ``[(dict, "gen_fixed0000")]``  (anchor_gen.py:1034-1084: encoder pass, sample_latents, decode, K-fold
                               ``"{k}_sample {i}"`` / ``"sample prior {i}"`` keys under cIMLE)
* eval, not ``gen`` (cIMLE) -> ``[(dict, "sample")]``         (:1085-1134: sample_noise, encode, decode — the reconstruction mode)
* train()                   -> loss dict                      (:1002-1021: prior_loss / fit_loss / mse_loss; stage 1 natively, see
                               ``training.stage1_losses``)

Everything heavy runs in libdfx through the encoder / diffusion mirrors; this file is the reference's bookkeeping (dict keys,
K-fold regrouping, which random draw happens where).  Random draws happen at the reference's sites in the reference's order
(reparameterisation -> latents -> aligner noise -> chain -> priors); the chain's noise is libdfx's Philox stream keyed by a seed
that is drawn from torch's generator at the point where the reference draws x_T (``engine.resolve_seed``).
Options outside the shipped ``configs/gen_*.py`` / ``train_*.py`` raise ``NotImplementedError``.
"""
import numpy as np
import torch
import torch.nn as nn

from . import modules as _modules
from .encoders import PartEncoderForTransformerDecoder
from .modules import AnchoredDiffusion


def _unsupported(what):
    raise NotImplementedError(f"libdfx implements the shipped gen_* / train_* AnchorDiffAE configuration only: {what}")


class Uniform:
    """``SAMPLERS['Uniform']`` (samplers/sampler.py:25-47): timesteps drawn with ``np.random.choice`` on the host, unit weights."""

    def __init__(self, num_timesteps):
        self.num_timesteps = int(num_timesteps)
        self.weight = np.ones([self.num_timesteps])

    def weights(self):
        return self.weight

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        return torch.from_numpy(idx).long().to(device), torch.from_numpy(1 / (len(p) * p[idx])).float().to(device)


def _fold(v, h):
    """einops ``rearrange(v, "(b h) ... -> b h ...", h=h)`` (anchor_gen.py:1062)."""
    return v.reshape(v.shape[0] // h, h, *v.shape[1:])


class AnchorDiffAE(nn.Module):
    def __init__(self, encoder, diffusion, sampler, num_anchors, num_timesteps, npoints=2048, zero_anchors=False, gen=False,
                 sample_noise_num=20, cimle=False, cimle_sample_num=10, diffusion_loss_weight=1.0, use_input=False, learn_var=False,
                 detach_variance=True, detach_anchor=True, global_shift=False, global_scale=False, vertical_only=True, ret_traj=False,
                 ret_interval=20, forward_sample=False, interpolate=False, interpolate_part_id=2, fix_part_ids=None, combine=False,
                 drift_anchors=False, save_pred_xstart=False, save_dir=None, save_weights=False, noise_reg_loss=True,
                 reg_loss_weight=1.0, pretrain_prior=False, train_language=False, language_encoder=None, clip_weight=1.0,
                 triplet_weight=1.0, triplet_thresh=0.1, precision="bf16"):
        super().__init__()
        for name, val in (("zero_anchors", zero_anchors), ("use_input", use_input), ("interpolate", interpolate), ("combine", combine),
                          ("drift_anchors", drift_anchors), ("save_weights", save_weights), ("pretrain_prior", pretrain_prior),
                          ("train_language", train_language), ("forward_sample", forward_sample)):
            if val:
                _unsupported(f"{name}=True")
        if isinstance(encoder, nn.Module):
            self.encoder = encoder
        else:
            cfg = dict(encoder)
            if cfg.pop("type", "PartEncoderForTransformerDecoder") != "PartEncoderForTransformerDecoder":
                _unsupported("encoder type other than PartEncoderForTransformerDecoder")
            self.encoder = PartEncoderForTransformerDecoder(**cfg)
        if isinstance(diffusion, nn.Module):
            self.diffusion = diffusion
        else:
            cfg = dict(diffusion)
            if cfg.pop("type", "AnchoredDiffusion") != "AnchoredDiffusion":
                _unsupported("diffusion type other than AnchoredDiffusion")
            self.diffusion = AnchoredDiffusion(num_timesteps=num_timesteps, precision=precision, **cfg)   # anchor_gen.py:87
        if isinstance(sampler, dict):
            if sampler.get("type", "Uniform") != "Uniform":
                _unsupported("sampler type other than Uniform")
            sampler = Uniform(num_timesteps)
        self.sampler = sampler
        self.diffusion_loss_weight, self.sample_noise_num, self.cimle, self.cimle_sample_num = \
            diffusion_loss_weight, sample_noise_num, cimle, cimle_sample_num
        self.fix_part_ids, self.gen = fix_part_ids, gen
        self.num_timesteps, self.num_anchors, self.npoints = int(num_timesteps), num_anchors, npoints
        # detach_anchor=False raises in the training forward (stage1_losses); detach_variance detaches a tensor the reference no longer reads
        # (anchor_gen.py:1013-1014 vs :1002), and learn_var / global_shift / global_scale / vertical_only are stored and never read by the
        # reference (:95-102): accepted and without effect, like there; noise_reg_loss / reg_loss_weight only enter the language branches
        # (:891,:911), which raise above (train_language).
        self.detach_anchor, self.detach_variance = detach_anchor, detach_variance
        self.fixed_id = [0] * num_anchors
        self.points_per_anchor = npoints // num_anchors
        self.ret_traj, self.ret_interval, self.save_pred_xstart = ret_traj, ret_interval, save_pred_xstart

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, anchors, ctx=None, noise=None, variance=None, anchor_assignments=None, valid_id=None, device="cuda", seed=None,
               generator=None, x_T_noise=None, step_noise=None):
        """anchor_gen.py:145-169.  ``anchors`` / ``variance`` (B,3,N) are the gathers of ctx[1] by ``anchor_assignments`` on every
        call path of the reference (:1044-1045, :1093-1099) — the kernel indexes ctx[1] itself.  ``noise``: an explicit x_T point
        cloud (B,3,N) as in the reference (:153, anchored_diffusion.py:560-561); then, or with ``save_pred_xstart``, the chain is
        walked one launch per step, else it is ONE persistent launch.  ``x_T_noise`` (B,3,N standard normal) / ``step_noise``
        (T,B,3,N) / ``seed`` / ``generator`` are libdfx extras (parity replays, reproducible runs)."""
        return _modules.decode(self.diffusion, ctx, anchor_assignments, valid_id=valid_id, ret_traj=self.ret_traj,
                               ret_interval=self.ret_interval, seed=seed, generator=generator, x_T_noise=x_T_noise,
                               step_noise=step_noise, save_pred_xstart=self.save_pred_xstart, x_T=noise)

    def sample(self, sample_num, fixed_id, valid_id, device, epoch, K=10):
        """anchor_gen.py:798-801 (K is ignored there as well: cimle_sample_num rows per shape)."""
        return self.encoder.sample_latents(sample_num, self.npoints, device, fixed_id=torch.as_tensor(fixed_id).to(device),
                                           valid_id=valid_id, epoch=epoch, K=self.cimle_sample_num, part_code=None)

    @torch.no_grad()
    def cache_noise(self, pcds, device, eval_whole=False):
        """anchor_gen.py:807-815 (cIMLE noise caching of stage 2): the best of ``sample_noise_num`` aligner noises per shape."""
        if eval_whole:
            # (the reference's own eval_whole branch cannot run: it unpacks FIVE values from the encoder's six-tuple, anchor_gen.py:819 vs
            # part_encoders.py:1254 — ValueError before any arithmetic; no shipped config sets eval_whole: nothing to mirror)
            _unsupported("cache_noise(eval_whole=True) (the reference's branch raises ValueError at anchor_gen.py:819)")
        noise, idx = self.encoder.sample_noise(pcds, device, self.sample_noise_num)
        return noise[torch.arange(noise.shape[0], device=noise.device), idx]

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, pcds, device="cuda", epoch=0, **kwargs):
        """anchor_gen.py:970-1136."""
        inp = pcds["input"].to(device)
        ref = pcds["ref"].to(device)
        input_seg_mask = pcds["seg_mask"].to(device)
        seg_mask = pcds["ref_seg_mask"].to(device)
        valid_id = pcds.get("present", None)
        dp_valid_id = pcds.get("dp_present", None)
        valid_id = None if valid_id is None else valid_id.to(device)
        B, N, C = ref.shape
        if self.npoints < N:
            _unsupported("npoints smaller than the reference cloud (anchor_gen.py:997-1001)")
        if self.training:
            from . import training as _training
            t, _ = self.sampler.sample(B, device)
            return _training.stage1_losses(self.encoder, self.diffusion, pcds, device=device, epoch=epoch, t=t,
                                           diffusion_loss_weight=self.diffusion_loss_weight, detach_anchor=self.detach_anchor)
        with torch.no_grad():
            # the reference runs the encoder on every val batch, gen branch included (:995): its reparameterisation draw comes first
            ctx, mean_pp, logvar_pp, _flag, _losses, _latents = self.encoder(pcds, device, epoch=epoch)
            h = self.cimle_sample_num
            if self.gen:
                fixed_id = [0] * self.num_anchors
                for i in (self.fix_part_ids or ()):
                    fixed_id[i] = 1
                ctx, mean_pp, logvar_pp, _seg, _valid, _lat = self.sample(B, fixed_id, valid_id, device, epoch, K=10)
                var_pp = torch.exp(logvar_pp)
                _pred = self.decode(mean_pp, ctx=ctx, device=device, variance=var_pp, anchor_assignments=_seg.to(torch.int32), valid_id=_valid)
                priors = torch.randn_like(var_pp.transpose(1, 2)) * torch.sqrt(var_pp.transpose(1, 2)) + mean_pp.transpose(1, 2)
                if self.cimle:
                    pred = {}
                    for i in range(h):
                        for k, v in _pred.items():
                            pred[f"{k}_sample {i}"] = _fold(v, h)[:, i]
                    for i in range(h):
                        pred[f"sample prior {i}"] = priors.reshape(B, h, self.npoints, C)[:, i]
                    pred["pred"] = _fold(_pred["pred"], h)[:, 0]
                    pred["pred_seg_mask"] = _fold(_seg, h)[:, 0]
                    pred["anchors"] = _fold(mean_pp, h)[:, 0].transpose(1, 2)
                else:
                    pred = _pred
                    pred["sample prior"] = priors
                    pred["pred_seg_mask"] = _seg
                    pred["anchors"] = mean_pp.transpose(1, 2)
                pred.update({"input": inp, "input_ref": ref, "ref_seg_mask": pcds["ref_seg_mask"], "seg_mask": input_seg_mask,
                             "present": valid_id, "shift": pcds["shift"], "scale": pcds["scale"]})
                pred = {k: v.detach().cpu() for k, v in pred.items()}
                return [(pred, "gen_fixed" + "".join(str(i) for i in fixed_id))]
            # ---- reconstruction ("sample") mode :1085-1134 ----
            if self.cimle:
                noise, _ = self.encoder.sample_noise(pcds, device, h)
                ctx, mean_pp, logvar_pp, _, _, latents = self.encoder(pcds, device, noise=noise)
                part_code, mean, logvar, noise = latents
                seg_mask, valid_id = (t.repeat_interleave(h, dim=0) for t in (seg_mask, valid_id))
            var_pp = torch.exp(logvar_pp)
            Np = mean_pp.shape[-1]
            if self.npoints > Np:                                                               # :1091-1093
                mean_pp, var_pp = (t.repeat_interleave(self.npoints // Np, dim=-1) for t in (mean_pp, var_pp))
                seg_mask = seg_mask.repeat_interleave(self.npoints // Np, dim=-1)
            _pred = self.decode(mean_pp, ctx=ctx, device=device, variance=var_pp, anchor_assignments=seg_mask.to(torch.int32), valid_id=valid_id)
            if self.cimle:
                pred = {}
                for i in range(h):
                    for k, v in _pred.items():
                        pred[f"{k}_sample {i}"] = _fold(v, h)[:, i]
                for i in range(h):
                    priors = torch.randn_like(var_pp).transpose(1, 2) * torch.sqrt(var_pp.transpose(1, 2)) + mean_pp.transpose(1, 2)
                    pred[f"sample prior {i}"] = priors.reshape(B, h, self.npoints, C)[:, i]
                    pred[f"noise latent {i}"] = noise.reshape(B, h, -1)[:, i]
                    pred[f"sample {i} mean"] = mean.reshape(B, h, 3, self.num_anchors)[:, i]
                    pred[f"sample {i} logvar"] = logvar.reshape(B, h, 3, self.num_anchors)[:, i]
                pred["pred"] = _fold(_pred["pred"], h)[:, 0]
                pred["pred_seg_mask"] = _fold(seg_mask, h)[:, 0]
                pred["anchors"] = _fold(mean_pp, h)[:, 0].transpose(1, 2)
                pred["part_latents"] = _fold(part_code, h)[:, 0]
                pred["valid_id"] = _fold(valid_id, h)[:, 0]
            else:
                pred = _pred
                pred["pred_seg_mask"] = seg_mask
                pred["anchors"] = mean_pp.transpose(1, 2)
                pred["sample prior"] = torch.randn_like(var_pp.transpose(1, 2)) * torch.sqrt(var_pp.transpose(1, 2)) + mean_pp.transpose(1, 2)
            pred.update({"input": inp, "input_ref": ref, "ref_seg_mask": pcds["ref_seg_mask"], "seg_mask": input_seg_mask,
                         "token": pcds["token"], "present": valid_id, "shift": pcds["shift"], "scale": pcds["scale"]})
            pred = {k: v.detach().cpu() if isinstance(v, torch.Tensor) else v for k, v in pred.items()}
            return [(pred, "sample")]

"""Host-side mirrors of the reference module API for the sampling hot path, backed by libdfx.

* ``TransformerNet``      same constructor arguments, parameter names and ``forward`` signature as
                          ``NETS['TransformerNet']`` (python/difffacto/models/diffusions/nets/attention.py:308-440);
                          the forward pass is ONE launch of the fused HIP denoiser.
* ``AnchoredDiffusion``   same constructor arguments and sampling methods as ``DIFFUSIONS['AnchoredDiffusion']``
                          (python/difffacto/models/diffusions/anchored_diffusion.py:13-588): the generator
                          ``p_sample_loop_progressive`` (one launch per step) and the fused ``sample_chain``.
* ``decode``              ``AnchorDiffAE.decode`` (python/difffacto/models/networks/anchor_gen.py:145-169) on top of
                          the single-launch persistent chain.

``state_dict`` keys are the reference's (``pretrained/*.pth`` loads with ``load_state_dict``).  Only the option
set of the shipped ``configs/gen_*.py`` / ``train_*.py`` is implemented natively; any other combination raises
``NotImplementedError`` (there is no PyTorch fallback path).
"""
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import training as _training
from .engine import DenoiserEngine, resolve_seed


# ---- parameter containers with the reference's attribute names (no torch math in them) -----------------------
class _GEGLUParams(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)          # attention.py:50-57


class _FeedForwardParams(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.Sequential(_GEGLUParams(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim))  # :77-94


class _CrossAttentionParams(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)  # attention.py:170-177
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class _BlockParams(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim, dropout=0.0):
        super().__init__()
        self.ff = _FeedForwardParams(dim, dropout=dropout)   # attention.py:279-283 (single_attn: no attn1/norm1)
        self.attn2 = _CrossAttentionParams(dim, context_dim, n_heads, d_head, dropout)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


def _unsupported(what):
    raise NotImplementedError(f"libdfx implements the shipped gen_*/train_* denoiser configuration only: {what}")


class TransformerNet(nn.Module):
    """Drop-in for ``NETS['TransformerNet']`` (attention.py:308-440)."""

    def __init__(self, in_channels, n_heads, d_head, out_channels, depth=1, dropout=0., context_dim=None,
                 use_linear=False, use_checkpoint=False, single_attn=False, class_cond=False, n_class=4,
                 cat_params_to_x=False, mask_out_unreferenced_code=True, cat_class_to_x=False,
                 use_sine_proj_in=False, add_t_to_x=False, res=False, add_class_cond=False, context_proj=False,
                 include_std=False):
        super().__init__()
        if not (use_linear and single_attn and class_cond and cat_params_to_x and cat_class_to_x
                and mask_out_unreferenced_code):
            _unsupported("use_linear, single_attn, class_cond, cat_params_to_x, cat_class_to_x must be True")
        if use_sine_proj_in or add_t_to_x or res or add_class_cond or context_proj or include_std:
            _unsupported("use_sine_proj_in / add_t_to_x / res / add_class_cond / context_proj / include_std")
        if (in_channels, out_channels, n_heads, d_head, n_class, context_dim) != (3, 3, 8, 16, 4, 262):
            _unsupported("in/out channels 3, 8 heads x 16, n_class 4, context_dim 256+6")
        self.n_class = n_class
        self.in_channels = in_channels + 6 + n_class          # attention.py:330
        self.inner_dim = inner = n_heads * d_head
        self.context_dim = context_dim + 256 + n_class        # attention.py:335
        self.depth = depth
        self.pre_norm = nn.LayerNorm(inner)
        self.post_norm = nn.LayerNorm(inner)
        self.proj_in = nn.Linear(self.in_channels, inner)
        self.time_embed = _FeedForwardParams(256, dropout=dropout)
        self.transformer_blocks = nn.ModuleList(
            [_BlockParams(inner, n_heads, d_head, self.context_dim, dropout) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, out_channels)        # in != out and not res -> not zero-initialised (:382)
        # engine cache (frozen weights): rebuilt when any parameter is modified in place or replaced
        self._dfx_T = 1000
        self._dfx_betas = (1e-4, 0.02)
        self._dfx_precision = "bf16"
        self._engine = None
        self._engine_key = None
        self._ctx_cache = None

    # -- libdfx plumbing --
    def configure(self, num_timesteps=None, beta_1=None, beta_T=None, precision=None):
        if num_timesteps is not None:
            self._dfx_T = int(num_timesteps)
        if beta_1 is not None and beta_T is not None:
            self._dfx_betas = (float(beta_1), float(beta_T))
        if precision is not None:
            self._dfx_precision = precision
        self._engine = None
        return self

    def engine(self):
        params = dict(self.named_parameters())
        # training.Adam updates parameters through raw pointers (no version bump): the generation of THIS module's parameters is part of the key
        key = (self._dfx_T, self._dfx_betas, self._dfx_precision, _training.generation_of(params.values()),
               tuple((p.data_ptr(), p._version) for p in params.values()))
        if self._engine is None or key != self._engine_key:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("TransformerNet (libdfx) needs its parameters on a HIP device: CPU not supported")
            self._engine = DenoiserEngine({k: v.detach() for k, v in params.items()}, self._dfx_T, *self._dfx_betas,
                                          precision=self._dfx_precision, device=dev)
            self._engine_key = key
            self._ctx_cache = None
        return self._engine

    def shape_context(self, ctx, valid_id):
        """ctx = [part_code (B,256,4), cat(mean, var) (B,6,4)] as built by prepare_ctx (part_encoders.py:1317-1326)."""
        if not isinstance(ctx, (list, tuple)) or len(ctx) != 2:
            _unsupported("ctx must be the [part_code, params] list of PartEncoderForTransformerDecoder.prepare_ctx")
        pc, params = ctx
        if valid_id is None:
            valid_id = torch.ones(pc.shape[0], self.n_class, device=pc.device)
        eng = self.engine()
        # The sampling loop passes the SAME tensor objects for all T steps: cache on object identity (weak
        # references, so a recycled allocation can never alias a dead tensor) + in-place version counters.
        c = self._ctx_cache
        if c is not None and c[0]() is pc and c[1]() is params and c[2]() is valid_id \
                and c[3] == (pc._version, params._version, valid_id._version):
            return c[4]
        sc = eng.prepare_shapes(pc, params[:, :3], params[:, 3:], valid_id.to(torch.float32))
        self._ctx_cache = (weakref.ref(pc), weakref.ref(params), weakref.ref(valid_id),
                           (pc._version, params._version, valid_id._version), sc)
        return sc

    def forward(self, x, t, ctx, anchors=None, variances=None, valid_id=None, anchor_assignment=None, **kwargs):
        """eps = eps_theta(x (B,3,N), t (B,), ctx).  ``anchors`` / ``variances`` (B,N,3) are accepted for signature
        compatibility; on the reference's call path (anchored_diffusion.py:261) they are the gather of ctx[1] by
        ``anchor_assignment``, which the kernel performs itself."""
        if anchor_assignment is None:
            _unsupported("anchor_assignment is required (cat_class_to_x)")
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())
                                        or any(c.requires_grad for c in ctx)):
            return self._forward_train(x, t, ctx, anchors, variances, valid_id, anchor_assignment)
        tt = t if isinstance(t, int) else int(t.reshape(-1)[0].item())
        if not isinstance(t, int) and t.numel() > 1 and not bool((t == t.reshape(-1)[0]).all()):
            _unsupported("per-sample timesteps (the sampling loop uses one t for the batch, anchored_diffusion.py:576)")
        sc = self.shape_context(ctx, valid_id)
        return self.engine().eps(sc, x, anchor_assignment, tt)

    def _forward_train(self, x, t, ctx, anchors, variances, valid_id, anchor_assignment):
        """Differentiable evaluation (per-shape t, saved activations, exact fp32): libdfx's training kernels behind
        torch.autograd (difffacto_amd/training.py).  In train() mode the modules' Dropout(p) is applied with libdfx's Philox
        factors (p = 0 is the parity setting of SURVEY.md §7 config 5)."""
        if next(self.parameters()).device.type != "cuda" or x.device.type != "cuda":
            raise RuntimeError("TransformerNet (libdfx) needs its parameters and inputs on a HIP device: CPU not supported")
        dropout = None
        if self.training:
            ps = {float(m.p) for m in self.modules() if isinstance(m, nn.Dropout)}
            if len(ps) > 1:
                _unsupported("different dropout probabilities inside one TransformerNet")
            if ps and max(ps) > 0:
                # one seed per step from torch's CPU generator (reproducible under torch.manual_seed); libdfx's own Philox
                # contract: torch's CUDA dropout stream depends on its launch geometry and cannot be reproduced
                dropout = (max(ps), int(torch.randint(0, 2 ** 62, ()).item()))
        if not isinstance(ctx, (list, tuple)) or len(ctx) != 2:
            _unsupported("ctx must be the [part_code, params] list of PartEncoderForTransformerDecoder.prepare_ctx")
        if anchors is None or variances is None:
            # the caller's per-point anchors / variances are the gather of ctx[1] by anchor_assignment
            # (anchored_diffusion.py:261, part_encoders.py:417-428); they are data, not differentiated
            idx = anchor_assignment.long()[:, None, :].expand(-1, 3, -1)
            par = ctx[1].detach()
            anchors = torch.gather(par[:, :3], 2, idx).transpose(1, 2).contiguous()
            variances = torch.gather(par[:, 3:], 2, idx).transpose(1, 2).contiguous()
        B = x.shape[0]
        tt = torch.full((B,), int(t), device=x.device) if isinstance(t, int) else t.reshape(-1).expand(B) if t.numel() == 1 else t
        return _training.denoiser_train_forward(dict(self.named_parameters()), x, tt, ctx[0], ctx[1], anchors, variances,
                                                valid_id, anchor_assignment, precision=self._dfx_precision, dropout=dropout)


class AnchoredDiffusion(nn.Module):
    """Drop-in for ``DIFFUSIONS['AnchoredDiffusion']`` (anchored_diffusion.py:13-588), sampling side."""

    def __init__(self, net, num_timesteps, beta_1, beta_T, k=1., res=True, mode='linear', use_beta=True,
                 rescale_timesteps=False, loss_type='mse', model_mean_type='epsilon', model_var_type='fixed_small',
                 scale_loss=False, clip_xstart=False, include_anchors=True, include_cov=False, learn_anchor=True,
                 learn_variance=False, classifier_weight=1., guidance=False, ddim_sampling=False, ddim_nsteps=10,
                 ddim_discretize='uniform', ddim_eta=1., precision="bf16"):
        super().__init__()
        if (mode != 'linear' or res or use_beta or rescale_timesteps or model_mean_type != 'epsilon'
                or model_var_type != 'fixed_small' or clip_xstart or include_anchors or include_cov
                or not learn_anchor or not learn_variance or guidance or scale_loss or loss_type != 'mse'):
            _unsupported("AnchoredDiffusion options other than those of configs/gen_*.py")
        if isinstance(net, nn.Module):
            self.model = net
        else:
            args = dict(net)
            args.pop("type", None)
            self.model = TransformerNet(**args)
        self.num_timesteps = int(num_timesteps)
        self.beta_1, self.beta_T = beta_1, beta_T
        self.model.configure(self.num_timesteps, beta_1, beta_T, precision)
        # schedule tables, float64 numpy like the reference (:62-112); the kernels use libdfx's own copy, which
        # tests pin bit-exactly against the reference
        betas = np.linspace(beta_1, beta_T, num=self.num_timesteps, dtype=np.float64)
        self.betas = betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.ddim_sampling = bool(ddim_sampling)
        if self.ddim_sampling:   # :114-124
            import math
            self.ddim_eta = ddim_eta
            if ddim_discretize == 'uniform':
                self.steps = list(range(0, self.num_timesteps, self.num_timesteps // ddim_nsteps))
            elif ddim_discretize == 'quad':
                self.steps = (np.linspace(0., math.sqrt(self.num_timesteps * 0.8), ddim_nsteps) ** 2).astype(np.int32).tolist()
            else:
                _unsupported(f"ddim_discretize={ddim_discretize!r}")
        else:
            self.steps = list(range(self.num_timesteps))

    # ---- sampling ----
    def _sc(self, ctx, valid_id):
        return self.model.shape_context(ctx, valid_id)

    @torch.no_grad()
    def p_sample(self, x, t, anchors, ctx=None, variance=None, anchor_assignment=None, valid_id=None, noise=None,
                 seed=None, generator=None):
        """anchored_diffusion.py:450-484.  Returns {'sample', 'pred_xstart'} like the reference.  The step's noise z is the
        in-kernel Philox stream keyed by (seed, point, t); ``seed=None`` draws a fresh key per call from ``generator`` / torch's
        global generator (``engine.resolve_seed``), as the reference draws a fresh ``randn_like`` (:476)."""
        tt = t if isinstance(t, int) else int(t.reshape(-1)[0].item())
        if self.ddim_sampling:
            sample, xs = self.model.engine().p_sample_ddim(self._sc(ctx, valid_id), x, anchor_assignment, tt, self.ddim_eta,
                                                           noise=noise, seed=seed, want_xstart=True, generator=generator)
            return {"sample": sample, "pred_xstart": xs}
        sample, xs = self.model.engine().p_sample(self._sc(ctx, valid_id), x, anchor_assignment, tt, noise=noise,
                                                  seed=seed, want_xstart=True, generator=generator)
        return {"sample": sample, "pred_xstart": xs}

    @torch.no_grad()
    def p_sample_loop_progressive(self, shape, anchors, ctx=None, variance=None, anchor_assignment=None,
                                  valid_id=None, noise=None, device=None, progress=False, seed=None, generator=None):
        """Generator of (t, {'sample': ...}) with the reference's protocol (:528-588): first (T, x_T), then one
        p_sample per step.  One kernel launch per step; use ``sample_chain`` for the single-launch path.
        ``seed=None`` (what the reference's ``decode`` passes, anchor_gen.py:149-158): ONE fresh key for this loop, drawn from
        ``generator`` / torch's global generator — every call sees new noise, ``torch.manual_seed`` replays it."""
        B, _, N = shape
        seed = resolve_seed(seed, generator)
        if noise is not None:
            pcd = noise
        else:
            L = torch.sqrt(variance)
            g = torch.Generator(device=variance.device)
            g.manual_seed(seed)
            pcd = L * torch.randn(*shape, device=variance.device, generator=g) + anchors
        yield self.num_timesteps, dict(sample=pcd)
        for i in self.steps[::-1]:
            out = self.p_sample(pcd, i, anchors, ctx=ctx, variance=variance, anchor_assignment=anchor_assignment,
                                valid_id=valid_id, seed=seed)
            yield i, out
            pcd = out["sample"]

    @torch.no_grad()
    def p_sample_loop(self, shape, anchors, ctx=None, noise=None, variance=None, anchor_assignment=None,
                      valid_id=None, device=None, progress=False, seed=None, generator=None):
        """:486-526 — final sample only; runs the fused chain when no explicit x_T is given."""
        seed = resolve_seed(seed, generator)
        if noise is None:
            pred, _ = self.sample_chain(ctx, anchor_assignment, valid_id, seed=seed)
            return pred.transpose(1, 2).contiguous()
        final = None
        for _, sample in self.p_sample_loop_progressive(shape, anchors, ctx=ctx, variance=variance, noise=noise,
                                                        anchor_assignment=anchor_assignment, valid_id=valid_id, seed=seed):
            final = sample
        return final["sample"]

    @torch.no_grad()
    def sample_chain(self, ctx, anchor_assignment, valid_id=None, x_T_noise=None, step_noise=None, seed=None,
                     ret_interval=None, shape_offset=0, generator=None):
        """Whole reverse chain in ONE persistent launch: (pred (B,N,3), traj (n_keep,B,N,3) | None).  `shape_offset`: global
        index of shape 0 (a rank of a sharded run passes its first shape's index: same clouds for any GPU count).
        ``seed=None``: a fresh Philox key per call (``engine.resolve_seed``); sharded runs pass ONE explicit seed to all ranks."""
        if self.ddim_sampling:
            return self.model.engine().sample_chain_ddim(self._sc(ctx, valid_id), anchor_assignment, self.steps, self.ddim_eta,
                                                         x_T_noise=x_T_noise, step_noise=step_noise, seed=seed,
                                                         ret_interval=ret_interval, shape_offset=shape_offset, generator=generator)
        return self.model.engine().sample_chain(self._sc(ctx, valid_id), anchor_assignment, x_T_noise=x_T_noise,
                                                step_noise=step_noise, seed=seed, ret_interval=ret_interval,
                                                shape_offset=shape_offset, generator=generator)

    def training_losses(self, x_start, t, anchors=None, variance=None, ctx=None, reduce=True, anchor_assignment=None,
                        valid_id=None, flags=None, noise=None):
        """anchored_diffusion.py:760-853: {'mse_loss'} of the epsilon objective at per-shape timesteps ``t`` (B,)
        (q_sample -> denoiser -> masked MSE).  Under ``no_grad`` in ``eval()`` the value comes from the inference
        engine; with gradients enabled it is differentiable in the denoiser's parameters, in ``ctx`` and in ``variance`` through libdfx's
        training kernels (Dropout(p) of train() mode included; anchors are data: the reference's agent detaches them, anchor_gen.py:1011-1012)."""
        if not reduce:
            _unsupported("training_losses(reduce=False)")
        if noise is None:
            noise = torch.randn_like(x_start)
        if self.training or torch.is_grad_enabled():
            if anchors is None or variance is None:
                _unsupported("anchors / variance (B,3,N) are required on the training path")
            if anchors.requires_grad:
                _unsupported("a gradient through the anchors (AnchorDiffAE detaches them: detach_anchor=True, anchor_gen.py:1011-1012)")
            # Like the reference, nothing is detached HERE (:779-791).  `variance` may carry a gradient: the reference's agent computes it from
            # logvar_per_point BEFORE its detach_variance line (anchor_gen.py:1002 vs :1013-1014), so in stage 2 the loss reaches the part aligner
            # through q_sample's sqrt(variance) * noise and through the network's per-point variance columns as well — reproduced.
            if variance.requires_grad:
                tab_a, tab_1 = self._q_tables(x_start.device)
                sa, s1 = tab_a[t].view(-1, 1, 1), tab_1[t].view(-1, 1, 1)
                x_t = sa * (x_start - anchors) + anchors + s1 * torch.sqrt(variance) * noise          # q_sample (:148-173), differentiable
            else:
                x_t = self.q_sample(x_start, t, anchors, noise=noise, variance=variance)
            eps = self.model(x_t, t, ctx, anchors=anchors.transpose(1, 2).contiguous(), variances=variance.transpose(1, 2).contiguous(),
                             valid_id=valid_id, anchor_assignment=anchor_assignment)
            return {"mse_loss": _training.masked_mse(noise, eps, flags)}
        eng = self.model.engine()
        sc = self._sc(ctx, valid_id)
        x_t = eng.q_sample(sc, anchor_assignment, x_start, t, noise)
        eps = eng.eps_t(sc, x_t, anchor_assignment, t)
        fl = None if flags is None else flags.reshape(flags.shape[0], -1)
        return {"mse_loss": eng.masked_mse(noise, eps, fl)}

    def _q_tables(self, device):
        """sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod) as fp32 device tensors, uploaded once per device (not per training step: a pageable
        host-to-device copy in the middle of a step drains the stream) and kept out of ``state_dict`` (the reference has no such buffers)."""
        cache = self.__dict__.setdefault("_q_tables_cache", {})
        key = str(device)
        if key not in cache:
            cache[key] = tuple(torch.from_numpy(np.asarray(a)).to(device).float() for a in (self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod))
        return cache[key]

    @torch.no_grad()
    def q_sample(self, x_start, t, anchors, noise=None, variance=None):
        """Forward process (:148-173), host-side elementwise helper (not on the sampling path)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        tab_a, tab_1 = self._q_tables(x_start.device)
        sa, s1 = tab_a[t].view(-1, 1, 1), tab_1[t].view(-1, 1, 1)
        return sa * (x_start - anchors) + anchors + s1 * torch.sqrt(variance) * noise


@torch.no_grad()
def decode(diffusion, ctx, anchor_assignments, valid_id=None, ret_traj=False, ret_interval=20, seed=None,
           x_T_noise=None, step_noise=None, shape_offset=0, generator=None, save_pred_xstart=False, x_T=None):
    """``AnchorDiffAE.decode`` (anchor_gen.py:145-169): {'pred': (B,N,3), t: (B,N,3) for t % ret_interval == 0}.
    ``seed=None`` (the default, and what the reference's call amounts to): fresh noise per call, replayable with
    ``torch.manual_seed``.  ``save_pred_xstart`` (:160-167) adds 'pred_xstart' / 'pred_xstart_{t}': those are outputs of every
    step, so that mode walks the chain one ``dfx_p_sample`` launch per step (the reference's own loop) instead of the
    single persistent launch.  ``x_T`` (B,3,N): an explicit starting CLOUD (the reference's ``noise`` argument of
    p_sample_loop_progressive, anchored_diffusion.py:560-561) instead of the draw ``x_T_noise``: stepwise as well."""
    if save_pred_xstart or x_T is not None:
        return _decode_stepwise(diffusion, ctx, anchor_assignments, valid_id, ret_traj, ret_interval, seed, x_T_noise, step_noise,
                                shape_offset, generator, save_pred_xstart, x_T)
    pred, traj = diffusion.sample_chain(ctx, anchor_assignments, valid_id, x_T_noise=x_T_noise, step_noise=step_noise,
                                        seed=seed, ret_interval=ret_interval if ret_traj else None, shape_offset=shape_offset,
                                        generator=generator)
    final = {"pred": pred}
    if ret_traj:
        visited = set(diffusion.steps) | {diffusion.num_timesteps}   # the prior sample t = T is always yielded (:565)
        for k, t in enumerate(diffusion.model.engine().snapshot_times(ret_interval)):
            if t in visited:
                final[t] = traj[k]
    return final


@torch.no_grad()
def _decode_stepwise(diffusion, ctx, seg, valid_id, ret_traj, ret_interval, seed, x_T_noise, step_noise, shape_offset, generator,
                     save_pred_xstart=True, x_T=None):
    """decode() with pred_xstart outputs: the reference's loop (anchor_gen.py:149-167) over ``dfx_p_sample`` launches.  x_T and every
    z_t are the explicit tensors when given, else the Philox streams of ONE key (x_T from a torch device generator seeded with it)."""
    seed = resolve_seed(seed, generator)
    eng = diffusion.model.engine()
    sc = diffusion._sc(ctx, valid_id)
    seg = seg.to(device=eng.device, dtype=torch.int32)
    B, N = seg.shape
    params = ctx[1].to(eng.device, torch.float32)
    idx = seg.long()[:, None, :].expand(-1, 3, -1)
    anchors, variance = torch.gather(params[:, :3], 2, idx), torch.gather(params[:, 3:], 2, idx)   # gather_all, part_encoders.py:417-428
    if x_T is not None:
        x = x_T.to(eng.device, torch.float32)                                                        # anchored_diffusion.py:560-561
    else:
        if x_T_noise is None:
            g = torch.Generator(device=eng.device)
            g.manual_seed(seed)
            x_T_noise = torch.randn(B, 3, N, device=eng.device, generator=g)
        x = torch.sqrt(variance) * x_T_noise.to(eng.device, torch.float32) + anchors                 # :563-564
    final = {}
    T = diffusion.num_timesteps
    if ret_traj and T % ret_interval == 0:
        final[T] = x.transpose(1, 2).contiguous()
    for k, t in enumerate(diffusion.steps[::-1]):
        z = None if step_noise is None else step_noise[k]
        if diffusion.ddim_sampling:
            x, xs = eng.p_sample_ddim(sc, x, seg, t, diffusion.ddim_eta, noise=z, seed=seed, want_xstart=True, shape_offset=shape_offset)
        else:
            x, xs = eng.p_sample(sc, x, seg, t, noise=z, seed=seed, want_xstart=True, shape_offset=shape_offset)
        if t == 0:
            final["pred"] = x.transpose(1, 2).contiguous()
            if save_pred_xstart:
                final["pred_xstart"] = xs.transpose(1, 2).contiguous()
        elif ret_traj and t % ret_interval == 0:
            final[t] = x.transpose(1, 2).contiguous()
            if save_pred_xstart:
                final[f"pred_xstart_{t}"] = xs.transpose(1, 2).contiguous()
    return final

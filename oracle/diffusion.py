"""numpy restatement of the anchored DDPM reverse process (ORACLE — test only).

Reference: python/difffacto/models/diffusions/anchored_diffusion.py
  * schedule tables                      :62-112  (float64 numpy, cast to fp32 at use by
                                          diffusion_utils.py:42-66 ``extract_into_tensor``)
  * ``p_mean_variance``                  :227-395 for the shipped config
        res=False, include_anchors=False, model_mean_type=EPSILON, model_var_type=FIXED_SMALL,
        learn_variance=True, learn_anchor=True, guidance=False, ddim_sampling=False
  * ``_predict_xstart_from_eps``         :401-409
  * ``q_posterior_mean`` / ``_variance`` :175-213
  * ``p_sample``                         :450-484
  * ``p_sample_loop_progressive``        :528-588
  * ``q_sample``                         :148-173
and ``AnchorDiffAE.decode`` python/difffacto/models/networks/anchor_gen.py:145-169.

Noise is always passed in explicitly (the reference draws ``torch.randn_like`` at :476 and
``torch.randn`` at :564; RNG streams cannot match across devices).
"""
import numpy as np

from . import denoiser as dn

F32 = np.float32


class Tables:
    """anchored_diffusion.py:62-112 (mode='linear')."""

    def __init__(self, num_timesteps, beta_1=1e-4, beta_T=0.02, ddim_sampling=False, ddim_nsteps=10,
                 ddim_discretize="uniform", ddim_eta=1.0):
        T = int(num_timesteps)
        self.T = T
        betas = np.linspace(beta_1, beta_T, num=T, dtype=np.float64)
        self.betas = betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef3 = 1.0 + ((np.sqrt(self.alphas_cumprod) - 1.0)
                                           * (np.sqrt(self.alphas_cumprod_prev) + np.sqrt(alphas))) / (1.0 - self.alphas_cumprod)
        self.ddim_sampling = bool(ddim_sampling)
        self.ddim_eta = float(ddim_eta)
        if ddim_sampling:   # :114-124
            self.xt_dir_coeff = np.sqrt(1.0 - self.alphas_cumprod - ddim_eta * ddim_eta * self.posterior_variance)
            if ddim_discretize == "uniform":
                self.steps = list(range(0, T, T // ddim_nsteps))
            elif ddim_discretize == "quad":
                self.steps = (np.linspace(0.0, np.sqrt(T * 0.8), ddim_nsteps) ** 2).astype(np.int32).tolist()
            else:
                raise NotImplementedError(ddim_discretize)
        else:
            self.steps = list(range(T))  # :126

    def f32(self, name, t):
        """extract_into_tensor: float64 table -> .float() -> index (diffusion_utils.py:42-66)."""
        return getattr(self, name).astype(F32)[t]


def predict_xstart_from_eps(tb, x_t, t, anchors, eps, sqrt_variance):
    """:401-409."""
    sra = tb.f32("sqrt_recip_alphas_cumprod", t)
    srm1 = tb.f32("sqrt_recipm1_alphas_cumprod", t)
    return (sra * (x_t - anchors) + anchors - srm1 * sqrt_variance * eps).astype(F32)


def q_posterior_mean(tb, x_start, x_t, t, anchors):
    """:175-193."""
    return (tb.f32("posterior_mean_coef1", t) * x_start
            + tb.f32("posterior_mean_coef2", t) * x_t
            + tb.f32("posterior_mean_coef3", t) * anchors).astype(F32)


def p_mean_variance(tb, W, x, t, anchors, ctx, variance, anchor_assignment, valid_id):
    """:227-395.  x, anchors, variance (B,3,N); t python int (same for the whole batch, :576)."""
    B = x.shape[0]
    tt = np.full((B,), t, dtype=np.int64)
    eps = dn.transformer_net_forward(W, x, tt, ctx, anchors.transpose(0, 2, 1), variance.transpose(0, 2, 1),
                                     valid_id, anchor_assignment)
    L = np.sqrt(variance).astype(F32)
    model_variance = (tb.f32("posterior_variance", t) * variance).astype(F32)        # FIXED_SMALL, :317-319
    pred_xstart = predict_xstart_from_eps(tb, x, t, anchors, eps, L)
    mean = q_posterior_mean(tb, pred_xstart, x, t, anchors)
    return dict(mean=mean, variance=model_variance, pred_xstart=pred_xstart, eps=eps)


def p_sample(tb, W, x, t, anchors, ctx, variance, anchor_assignment, valid_id, noise):
    """:450-484 — ``sample = mean + 1[t!=0] * sqrt(variance) * noise``; with ddim_sampling (:368-377, :480-481)
    ``(x0 - a) sqrt(acp_prev[t]) + a + L xt_dir_coeff[t] eps + eta 1[t!=0] sqrt(variance) noise``."""
    out = p_mean_variance(tb, W, x, t, anchors, ctx, variance, anchor_assignment, valid_id)
    nz = F32(1.0 if t != 0 else 0.0)
    if tb.ddim_sampling:
        L = np.sqrt(variance).astype(F32)
        xt_dir = (L * tb.f32("xt_dir_coeff", t) * out["eps"]).astype(F32)
        sample = ((out["pred_xstart"] - anchors) * np.sqrt(tb.f32("alphas_cumprod_prev", t)).astype(F32) + anchors + xt_dir
                  + F32(tb.ddim_eta) * nz * np.sqrt(out["variance"]).astype(F32) * noise).astype(F32)
    else:
        sample = (out["mean"] + nz * np.sqrt(out["variance"]).astype(F32) * noise).astype(F32)
    return dict(sample=sample, pred_xstart=out["pred_xstart"], eps=out["eps"])


def p_sample_loop_progressive(tb, W, anchors, ctx, variance, anchor_assignment, valid_id, x_T_noise, step_noise):
    """:528-588.  ``x_T_noise`` (B,3,N) is the randn of :564; ``step_noise[i]`` is the
    randn_like drawn inside the i-th executed p_sample call (i = 0 for t = T-1), shape (len(steps),B,3,N)."""
    pcd = (np.sqrt(variance).astype(F32) * x_T_noise + anchors).astype(F32)
    yield tb.T, dict(sample=pcd)
    for n, i in enumerate(tb.steps[::-1]):
        out = p_sample(tb, W, pcd, i, anchors, ctx, variance, anchor_assignment, valid_id, step_noise[n])
        yield i, out
        pcd = out["sample"]


def decode(tb, W, anchors, ctx, variance, anchor_assignment, valid_id, x_T_noise, step_noise,
           ret_traj=True, ret_interval=10):
    """AnchorDiffAE.decode (anchor_gen.py:145-169): keep t==0 as 'pred' (B,N,3) and every
    ``ret_interval``-th t (including t=T, the prior sample) when ``ret_traj``."""
    final = {}
    for t, sample in p_sample_loop_progressive(tb, W, anchors, ctx, variance, anchor_assignment, valid_id,
                                               x_T_noise, step_noise):
        if t == 0:
            final["pred"] = sample["sample"].transpose(0, 2, 1)
        elif ret_traj and t % ret_interval == 0:
            final[t] = sample["sample"].transpose(0, 2, 1)
    return final


def q_sample(tb, x_start, t, anchors, noise, variance):
    """:148-173 (training forward process; t is a per-shape int array (B,))."""
    L = np.sqrt(variance).astype(F32)
    sa = tb.f32("sqrt_alphas_cumprod", t)[:, None, None]
    s1 = tb.f32("sqrt_one_minus_alphas_cumprod", t)[:, None, None]
    return (sa * (x_start - anchors) + anchors + s1 * L * noise).astype(F32)


def gather_params(seg, mean, var):
    """PartEncoder.gather_all (part_encoders.py:417-428) via gather_operation semantics:
    per-point (B,3,N) params from per-part (B,3,J) params and ``seg`` (B,N) int."""
    idx = np.asarray(seg).astype(np.int64)[:, None, :]
    a = np.take_along_axis(mean, np.broadcast_to(idx, (mean.shape[0], 3, idx.shape[2])), axis=2)
    v = np.take_along_axis(var, np.broadcast_to(idx, (var.shape[0], 3, idx.shape[2])), axis=2)
    return a.astype(F32), v.astype(F32)


def training_losses(tb, W, x_start, t, anchors, variance, ctx, anchor_assignment, valid_id, flags, noise):
    """:760-853 in eval mode (no dropout) for the shipped config (EPSILON target, fixed_small, reduce=True):
    x_t = q_sample(x0, t); eps_hat = model(x_t, t); loss = ((noise - eps_hat)^2 * flags).mean(1).sum() / flags.sum().
    ``t`` is a per-shape int array (B,); flags (B,1,N) or None."""
    x_t = q_sample(tb, x_start, t, anchors, noise, variance)
    eps = dn.transformer_net_forward(W, x_t, np.asarray(t, dtype=np.int64), ctx, anchors.transpose(0, 2, 1),
                                     variance.transpose(0, 2, 1), valid_id, anchor_assignment)
    d = ((noise - eps) ** 2).astype(F32)
    if flags is not None:
        d = (d * flags).astype(F32)
        loss = d.mean(axis=1, dtype=F32).sum(dtype=F32) / flags.sum(dtype=F32)
    else:
        loss = d.mean(dtype=F32)
    return dict(mse_loss=F32(loss), x_t=x_t, eps=eps)

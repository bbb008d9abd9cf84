"""TEST-ONLY stand-ins for the two libdfx handles the host mirrors own (engine.DenoiserEngine, latents.LatentSampler), backed by the numpy
oracle.  They let the dev container (no GPU) push the REFERENCE's own ``AnchorDiffAE.forward`` through ``difffacto_amd.install()`` +
``encoders.attach()`` and check the call protocol (arguments, generator / seed handling, result layouts) the reference actually uses.
Never imported by the product; the product path has no such fallback."""
import numpy as np
import torch

from oracle import diffusion as odf
from oracle import latents as ol


class _Ctx:
    def __init__(self, part_code, mean, var, valid):
        self.part_code, self.mean, self.var, self.valid = part_code, mean, var, valid
        self.B = part_code.shape[0]


class OracleEngine:
    """The subset of DenoiserEngine that modules.TransformerNet / AnchoredDiffusion / decode call."""
    replay_steps = None          # list of (B,3,N) arrays: consumed one per p_sample call instead of the seeded stream
    seeds_seen = []

    def __init__(self, params, num_timesteps, beta_1=1e-4, beta_T=0.02, precision="bf16", device=None):
        self.W = {k: v.detach().cpu().numpy() for k, v in params.items()}
        self.num_timesteps = int(num_timesteps)
        self.tb = odf.Tables(self.num_timesteps, beta_1, beta_T)
        self.device = torch.device("cpu")
        self.precision = precision

    def prepare_shapes(self, part_code, mean, var, valid):
        f = lambda t: t.detach().cpu().numpy().astype(np.float32)
        return _Ctx(f(part_code), f(mean), f(var), f(valid))

    def _operands(self, ctx, seg):
        seg = seg.detach().cpu().numpy()
        anchors, variance = odf.gather_params(seg, ctx.mean, ctx.var)
        return seg, anchors, variance, [ctx.part_code, np.concatenate([ctx.mean, ctx.var], 1)]

    @staticmethod
    def _z(seed, t, shape):
        return np.random.default_rng([int(seed) % (2 ** 63), int(t)]).standard_normal(shape).astype(np.float32)

    def p_sample(self, ctx, x, seg, t, noise=None, seed=None, want_xstart=False, shape_offset=0, generator=None):
        from difffacto_amd.engine import resolve_seed
        x = x.detach().cpu().numpy().astype(np.float32)
        if noise is not None:
            z = noise.detach().cpu().numpy()
        elif OracleEngine.replay_steps is not None:
            z = OracleEngine.replay_steps.pop(0)
        else:
            seed = resolve_seed(seed, generator)
            OracleEngine.seeds_seen.append(seed)
            z = self._z(seed, t, x.shape)
        seg, anchors, variance, cl = self._operands(ctx, seg)
        out = odf.p_sample(self.tb, self.W, x, int(t), anchors, cl, variance, seg, ctx.valid, z)
        s, xs = torch.from_numpy(out["sample"]), torch.from_numpy(out["pred_xstart"])
        return (s, xs) if want_xstart else s

    def sample_chain(self, ctx, seg, x_T_noise=None, step_noise=None, seed=None, ret_interval=None, shape_offset=0, generator=None):
        from difffacto_amd.engine import resolve_seed
        B, N = seg.shape
        T = self.num_timesteps
        if x_T_noise is None or step_noise is None:
            seed = resolve_seed(seed, generator)
            OracleEngine.seeds_seen.append(seed)
        xT = x_T_noise.detach().cpu().numpy() if x_T_noise is not None else self._z(seed, T, (B, 3, N))
        zs = step_noise.detach().cpu().numpy() if step_noise is not None else np.stack([self._z(seed, t, (B, 3, N)) for t in range(T - 1, -1, -1)])
        seg, anchors, variance, cl = self._operands(ctx, seg)
        dec = odf.decode(self.tb, self.W, anchors, cl, variance, seg, ctx.valid, xT, zs, ret_traj=bool(ret_interval), ret_interval=ret_interval or 1)
        traj = None
        if ret_interval:
            traj = torch.from_numpy(np.stack([dec[t] for t in self.snapshot_times(ret_interval)]))
        return torch.from_numpy(dec["pred"]), traj

    def snapshot_times(self, ret_interval):
        nk = self.num_timesteps // ret_interval
        return [(nk - k) * ret_interval for k in range(nk)]


class OracleLatentSampler:
    def __init__(self, params, n_class=4, zdim=256, n_heads=8, d_head=32, cimle=True, noise_dim=32, noise_scale=10.0, prior_var=1.0,
                 log_scale_var=0.0, device=None):
        self.W = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in params.items()}
        self.n_class, self.zdim, self.noise_dim, self.cimle = n_class, zdim, noise_dim, cimle
        self.noise_scale, self.prior_var, self.log_scale_var = float(noise_scale), float(prior_var), float(log_scale_var)
        self.heads = n_heads

    def part_aligner(self, part_code, valid_id, noise=None):
        f = lambda t: t.detach().cpu().numpy().astype(np.float32)
        m, lv = ol.part_aligner_forward(self.W, f(part_code), f(valid_id), f(noise), noise_scale=self.noise_scale, heads=self.heads)
        return torch.from_numpy(m), torch.from_numpy(lv)

    def sample_latents(self, w_noise, aligner_noise, valid_id, fixed_id=None, K=1, npoints=2048, part_code=None):
        f = lambda t: None if t is None else t.detach().cpu().numpy().astype(np.float32)
        out = ol.sample_latents(self.W, f(w_noise), f(aligner_noise), f(valid_id), [0] * self.n_class if fixed_id is None else list(fixed_id),
                                int(K), int(npoints), prior_var=self.prior_var, noise_scale=self.noise_scale, log_scale_var=self.log_scale_var,
                                part_code=f(part_code))
        t = torch.from_numpy
        return {"part_code": t(out["part_code"]), "valid_id": t(out["valid_id"]), "noise": t(np.ascontiguousarray(out["noise"])), "mean": t(out["mean"]),
                "logvar": t(out["logvar"]), "params": t(out["ctx"][1]), "seg_mask": t(out["seg_mask"]), "mean_per_point": t(out["mean_per_point"]),
                "logvar_per_point": t(out["logvar_per_point"])}

"""Host-side driver of libdfx's denoiser: owns the opaque ``dfx_denoiser`` handle and the per-batch
shape context.  PyTorch is used for device memory and the stream only.

    eng = DenoiserEngine(params, num_timesteps=1000, precision="bf16")
    ctx = eng.prepare_shapes(part_code, mean, var, valid)       # once per batch of shapes
    pred, traj = eng.sample_chain(ctx, seg, ret_interval=10)    # the whole reverse chain, one launch

``params`` maps the reference ``state_dict`` names relative to ``diffusion.model.``
 to fp32 CUDA tensors.
"""
import ctypes

import numpy as np
import torch

from . import _ffi

PRECISIONS = {"f32": _ffi.DFX_PREC_F32, "fp32": _ffi.DFX_PREC_F32, "bf16": _ffi.DFX_PREC_BF16}

TABLE_NAMES = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
               "posterior_mean_coef2", "posterior_mean_coef3", "posterior_variance", "sqrt_alphas_cumprod",
               "sqrt_one_minus_alphas_cumprod")

_BLOCK_FIELDS = {
    "norm2_w": "norm2.weight", "norm2_b": "norm2.bias", "to_q": "attn2.to_q.weight", "to_k": "attn2.to_k.weight",
    "to_v": "attn2.to_v.weight", "to_out_w": "attn2.to_out.0.weight", "to_out_b": "attn2.to_out.0.bias",
    "norm3_w": "norm3.weight", "norm3_b": "norm3.bias", "ff0_w": "ff.net.0.proj.weight",
    "ff0_b": "ff.net.0.proj.bias", "ff2_w": "ff.net.2.weight", "ff2_b": "ff.net.2.bias",
}
_TOP_FIELDS = {
    "proj_in_w": "proj_in.weight", "proj_in_b": "proj_in.bias", "pre_norm_w": "pre_norm.weight",
    "pre_norm_b": "pre_norm.bias", "post_norm_w": "post_norm.weight", "post_norm_b": "post_norm.bias",
    "proj_out_w": "proj_out.weight", "proj_out_b": "proj_out.bias", "te0_w": "time_embed.net.0.proj.weight",
    "te0_b": "time_embed.net.0.proj.bias", "te2_w": "time_embed.net.2.weight", "te2_b": "time_embed.net.2.bias",
}

EXPECTED_SHAPES = {
    "proj_in.weight": (128, 13), "proj_in.bias": (128,), "proj_out.weight": (3, 128), "proj_out.bias": (3,),
    "time_embed.net.0.proj.weight": (2048, 256), "time_embed.net.2.weight": (256, 1024),
    "attn2.to_q.weight": (128, 128), "attn2.to_k.weight": (128, 522), "attn2.to_v.weight": (128, 522),
    "attn2.to_out.0.weight": (128, 128), "ff.net.0.proj.weight": (1024, 128), "ff.net.2.weight": (128, 512),
}


def _dev_f32(t, name, device):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    t = t.detach().to(device=device, dtype=torch.float32).contiguous()
    for suffix, shape in EXPECTED_SHAPES.items():
        if name.endswith(suffix) and tuple(t.shape) != shape:
            raise RuntimeError(f"{name}: expected shape {shape}, got {tuple(t.shape)} "
                               "(libdfx is specialised for the shipped gen_* denoiser)")
    return t


def resolve_seed(seed=None, generator=None):
    """The 64-bit Philox key of one sampling call.  ``seed=None`` (the default everywhere above the C-ABI) draws it from
    ``generator`` or, without one, from torch's global CPU generator — so noise is FRESH per call, and governed by
    ``torch.manual_seed`` / the reference's ``set_random_seed`` (runner.py:39), like the ``torch.randn`` /
    ``randn_like`` draws it replaces (anchored_diffusion.py:476,564).  A device generator is honoured as well (one
    host sync).  An explicit integer is used as it is: the reproducible / sharded calls (parallel.py) pass one."""
    if seed is not None:
        return int(seed)
    if generator is not None and generator.device.type != "cpu":
        return int(torch.randint(0, 2 ** 62, (), device=generator.device, generator=generator).item())
    return int(torch.randint(0, 2 ** 62, (), generator=generator).item())


def last_kernel_variant():
    """Name of the denoiser kernel the most recent launch took (``dfx_last_kernel_variant``), e.g. ``"k_denoise_pipe<8>"``."""
    return _ffi.lib().dfx_last_kernel_variant().decode()


class ShapeContext:
    """Per-batch static operands living in one device buffer (dfx_shape_ctx_prepare)."""

    def __init__(self, buf, B):
        self.buf = buf
        self.B = B


class DenoiserEngine:
    def __init__(self, params, num_timesteps, beta_1=1e-4, beta_T=0.02, precision="bf16", device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DenoiserEngine needs a HIP device (there is no CPU path)")
        self.device = torch.device(device if device is not None else "cuda")
        self.precision = precision
        prec = PRECISIONS[precision]
        depth = 0
        while f"transformer_blocks.{depth}.norm2.weight" in params:
            depth += 1
        if not 1 <= depth <= _ffi.DFX_MAX_DEPTH:
            raise RuntimeError(f"unsupported depth {depth}")
        self.depth = depth
        self.num_timesteps = int(num_timesteps)
        keep = []
        w = _ffi.DenoiserWeights()
        w.depth = depth
        for field, key in _TOP_FIELDS.items():
            t = _dev_f32(params[key], key, self.device)
            keep.append(t)
            setattr(w, field, t.data_ptr())
        for b in range(depth):
            for field, key in _BLOCK_FIELDS.items():
                full = f"transformer_blocks.{b}.{key}"
                t = _dev_f32(params[full], full, self.device)
                keep.append(t)
                setattr(w.blk[b], field, t.data_ptr())
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_denoiser_create(ctypes.byref(handle), ctypes.byref(w), self.num_timesteps,
                                               float(beta_1), float(beta_T), prec, _ffi.current_stream())
        _ffi.check(rc, "dfx_denoiser_create")
        del keep  # create() synchronised the stream: parameters are no longer referenced
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().dfx_denoiser_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def w1_fold(self):
        """(folded, ratio): whether this bf16 engine carries b1' in a hidden channel's K slot of the packed W1 (``dfx_denoiser_w1_fold``) and that
        channel's outlier ratio max_r |W1[r,k] gamma3[k]| / mean_c |W1[r,c] gamma3[c]| (not folded — every channel above 8 —: plain pack, direct kernel)."""
        r = ctypes.c_float(0.0)
        return bool(_ffi.lib().dfx_denoiser_w1_fold(self._h, ctypes.byref(r))), float(r.value)

    def w1_fold_channel(self):
        """The hidden channel create() made the redundant one (127 unless it relabelled the channels around an outlier; -1: no fold)."""
        return int(_ffi.lib().dfx_debug_w1_fold_channel(self._h))

    # ------------------------------------------------------------------------------------------
    def tables(self):
        """The fp32 schedule tables the kernels use, dict name -> (T,) numpy."""
        out = np.empty((8, self.num_timesteps), dtype=np.float32)
        _ffi.check(_ffi.lib().dfx_denoiser_get_tables(self._h, out.ctypes.data_as(ctypes.c_void_p)), "get_tables")
        return dict(zip(TABLE_NAMES, out))

    def prepare_shapes(self, part_code, mean, var, valid):
        """part_code (B,256,4), mean (B,3,4), var (B,3,4) [= exp(logvar)], valid (B,4) -> ShapeContext."""
        f = lambda t: t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        part_code, mean, var, valid = f(part_code), f(mean), f(var), f(valid)
        B = part_code.shape[0]
        if tuple(part_code.shape) != (B, 256, 4) or tuple(mean.shape) != (B, 3, 4) or tuple(var.shape) != (B, 3, 4) \
                or tuple(valid.shape) != (B, 4):
            raise RuntimeError("prepare_shapes: expected part_code (B,256,4), mean/var (B,3,4), valid (B,4)")
        nbytes = _ffi.lib().dfx_shape_ctx_bytes(self._h, B)
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_shape_ctx_prepare(self._h, _ffi.ptr(part_code), _ffi.ptr(mean), _ffi.ptr(var),
                                                 _ffi.ptr(valid), _ffi.ptr(buf), B, _ffi.current_stream())
        _ffi.check(rc, "dfx_shape_ctx_prepare")
        return ShapeContext(buf, B)

    @staticmethod
    def _seg(seg, device):
        seg = seg.detach().to(device=device, dtype=torch.int32).contiguous()
        return seg

    # The kernels take N % 32 == 0 (one wavefront = 32 points of one shape); the reference takes any N.  Points are
    # independent in the denoiser and in the posterior update, so other sizes run padded to the next multiple of 32 --
    # zeros for coordinates / noise, the shape's first label for seg -- and the padding is cut off again: exact.
    @staticmethod
    def _pad(N):
        return (-int(N)) % 32

    @staticmethod
    def _pad_last(t, pad):
        return None if t is None else torch.nn.functional.pad(t, (0, pad))

    @staticmethod
    def _pad_seg(seg, pad):
        return torch.cat([seg, seg[:, :1].expand(-1, pad)], dim=1).contiguous()

    def eps(self, ctx, x, seg, t):
        """TransformerNet.forward: x (B,3,N), seg (B,N) -> eps (B,3,N)."""
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        seg = self._seg(seg, self.device)
        B, _, N = x.shape
        pad = self._pad(N)
        if pad:
            return self.eps(ctx, self._pad_last(x, pad), self._pad_seg(seg, pad), t)[..., :N].contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_denoise_eps(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(x), _ffi.ptr(seg), int(t),
                                           _ffi.ptr(out), B, N, _ffi.current_stream())
        _ffi.check(rc, "dfx_denoise_eps")
        return out

    def p_sample(self, ctx, x, seg, t, noise=None, seed=None, want_xstart=False, shape_offset=0, generator=None):
        seed = 0 if noise is not None and seed is None else resolve_seed(seed, generator)   # explicit noise: no draw
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        seg = self._seg(seg, self.device)
        B, _, N = x.shape
        if noise is not None:
            noise = noise.detach().to(device=self.device, dtype=torch.float32).contiguous()
            assert noise.shape == x.shape
        pad = self._pad(N)
        if pad:
            r = self.p_sample(ctx, self._pad_last(x, pad), self._pad_seg(seg, pad), t, self._pad_last(noise, pad), seed, want_xstart, shape_offset)
            return tuple(a[..., :N].contiguous() for a in r) if want_xstart else r[..., :N].contiguous()
        out = torch.empty_like(x)
        xs = torch.empty_like(x) if want_xstart else None
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_p_sample(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(x), _ffi.ptr(seg), int(t),
                                        _ffi.ptr(noise), int(seed), int(shape_offset), _ffi.ptr(out), _ffi.ptr(xs), B, N,
                                        _ffi.current_stream())
        _ffi.check(rc, "dfx_p_sample")
        return (out, xs) if want_xstart else out

    def sample_chain(self, ctx, seg, x_T_noise=None, step_noise=None, seed=None, ret_interval=None, shape_offset=0, generator=None):
        """Whole reverse chain in one launch.  Returns (pred (B,N,3), traj or None) where traj is
        (n_keep,B,N,3) with snapshot k <-> t = (T // ret_interval - k) * ret_interval.  `shape_offset` = global index of
        shape 0 of this call: the in-kernel Philox noise is keyed by the global point id, so a batch split over calls / ranks
        gives the clouds of the unsplit call.  ``seed=None``: a fresh key from ``generator`` / torch's global generator
        (``resolve_seed``); nothing is drawn when both noise tensors are given."""
        seed = 0 if (x_T_noise is not None and step_noise is not None and seed is None) else resolve_seed(seed, generator)
        seg = self._seg(seg, self.device)
        B, N = seg.shape
        T = self.num_timesteps
        f = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        x_T_noise, step_noise = f(x_T_noise), f(step_noise)
        if x_T_noise is not None:
            assert tuple(x_T_noise.shape) == (B, 3, N)
        if step_noise is not None:
            assert tuple(step_noise.shape) == (T, B, 3, N)
        pad = self._pad(N)
        if pad:
            pred, traj = self.sample_chain(ctx, self._pad_seg(seg, pad), self._pad_last(x_T_noise, pad), self._pad_last(step_noise, pad),
                                           seed, ret_interval, shape_offset)
            return pred[:, :N].contiguous(), None if traj is None else traj[:, :, :N].contiguous()
        pred = torch.empty(B, N, 3, dtype=torch.float32, device=self.device)
        traj = None
        ri = 0
        if ret_interval:
            ri = int(ret_interval)
            nk = _ffi.lib().dfx_chain_num_snapshots(T, ri)
            traj = torch.empty(nk, B, N, 3, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_sample_chain(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(seg), _ffi.ptr(x_T_noise),
                                            _ffi.ptr(step_noise), int(seed), int(shape_offset), ri, _ffi.ptr(traj), _ffi.ptr(pred), B, N,
                                            _ffi.current_stream())
        _ffi.check(rc, "dfx_sample_chain")
        return pred, traj

    # ---- training-style forward evaluation (no dropout, no gradients) ----
    def _tvec(self, t, B):
        t = torch.as_tensor(t).detach().to(device=self.device, dtype=torch.int32).reshape(-1).contiguous()
        assert t.numel() == B
        return t

    def q_sample(self, ctx, seg, x_start, t, noise):
        """anchored_diffusion.py:148-173 with per-shape t (B,)."""
        f = lambda a: a.detach().to(device=self.device, dtype=torch.float32).contiguous()
        x_start, noise, seg = f(x_start), f(noise), self._seg(seg, self.device)
        B, _, N = x_start.shape
        pad = self._pad(N)
        if pad:
            return self.q_sample(ctx, self._pad_seg(seg, pad), self._pad_last(x_start, pad), t, self._pad_last(noise, pad))[..., :N].contiguous()
        out = torch.empty_like(x_start)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_q_sample_f32(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(seg), _ffi.ptr(self._tvec(t, B)), _ffi.ptr(x_start),
                                            _ffi.ptr(noise), _ffi.ptr(out), B, N, _ffi.current_stream())
        _ffi.check(rc, "dfx_q_sample_f32")
        return out

    def eps_t(self, ctx, x, seg, t):
        """TransformerNet.forward with one timestep per shape: t (B,)."""
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        seg = self._seg(seg, self.device)
        B, _, N = x.shape
        pad = self._pad(N)
        if pad:
            return self.eps_t(ctx, self._pad_last(x, pad), self._pad_seg(seg, pad), t)[..., :N].contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_denoise_eps_t(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(x), _ffi.ptr(seg), _ffi.ptr(self._tvec(t, B)),
                                             _ffi.ptr(out), B, N, _ffi.current_stream())
        _ffi.check(rc, "dfx_denoise_eps_t")
        return out

    def masked_mse(self, target, pred, flags=None):
        """((target - pred)^2 * flags).mean(1).sum() / flags.sum() (anchored_diffusion.py:840-847) -> 0-dim tensor."""
        f = lambda a: None if a is None else a.detach().to(device=self.device, dtype=torch.float32).contiguous()
        target, pred, flags = f(target), f(pred), f(flags)
        B, _, N = target.shape
        ws = torch.empty(2, dtype=torch.float64, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_masked_mse_f32(_ffi.ptr(target), _ffi.ptr(pred), _ffi.ptr(flags), _ffi.ptr(ws), _ffi.ptr(loss), B, N,
                                              _ffi.current_stream())
        _ffi.check(rc, "dfx_masked_mse_f32")
        return loss[0]

    def p_sample_ddim(self, ctx, x, seg, t, eta, noise=None, seed=None, want_xstart=False, shape_offset=0, generator=None):
        """One DDIM update (anchored_diffusion.py:368-377, :480-481)."""
        seed = 0 if noise is not None and seed is None else resolve_seed(seed, generator)
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        seg = self._seg(seg, self.device)
        B, _, N = x.shape
        if noise is not None:
            noise = noise.detach().to(device=self.device, dtype=torch.float32).contiguous()
            assert noise.shape == x.shape
        pad = self._pad(N)
        if pad:
            r = self.p_sample_ddim(ctx, self._pad_last(x, pad), self._pad_seg(seg, pad), t, eta, self._pad_last(noise, pad), seed, want_xstart, shape_offset)
            return tuple(a[..., :N].contiguous() for a in r) if want_xstart else r[..., :N].contiguous()
        out = torch.empty_like(x)
        xs = torch.empty_like(x) if want_xstart else None
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_p_sample_ddim(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(x), _ffi.ptr(seg), int(t), float(eta),
                                             _ffi.ptr(noise), int(seed), int(shape_offset), _ffi.ptr(out), _ffi.ptr(xs), B, N,
                                             _ffi.current_stream())
        _ffi.check(rc, "dfx_p_sample_ddim")
        return (out, xs) if want_xstart else out

    def sample_chain_ddim(self, ctx, seg, steps, eta, x_T_noise=None, step_noise=None, seed=None, ret_interval=None, shape_offset=0,
                          generator=None):
        """DDIM chain in one launch over the ascending step list ``steps`` (executed in reverse).  Returns
        (pred, traj or None); traj slots follow ``snapshot_times``; only timesteps in ``steps`` are written."""
        seed = 0 if (x_T_noise is not None and step_noise is not None and seed is None) else resolve_seed(seed, generator)
        seg = self._seg(seg, self.device)
        B, N = seg.shape
        steps = [int(v) for v in steps]
        f = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        x_T_noise, step_noise = f(x_T_noise), f(step_noise)
        if step_noise is not None:
            assert tuple(step_noise.shape) == (len(steps), B, 3, N)
        pad = self._pad(N)
        if pad:
            pred, traj = self.sample_chain_ddim(ctx, self._pad_seg(seg, pad), steps, eta, self._pad_last(x_T_noise, pad),
                                                self._pad_last(step_noise, pad), seed, ret_interval, shape_offset)
            return pred[:, :N].contiguous(), None if traj is None else traj[:, :, :N].contiguous()
        pred = torch.empty(B, N, 3, dtype=torch.float32, device=self.device)
        traj, ri = None, 0
        if ret_interval:
            ri = int(ret_interval)
            traj = torch.zeros(_ffi.lib().dfx_chain_num_snapshots(self.num_timesteps, ri), B, N, 3, dtype=torch.float32,
                               device=self.device)
        arr = (ctypes.c_int32 * len(steps))(*steps)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_sample_chain_ddim(self._h, _ffi.ptr(ctx.buf), _ffi.ptr(seg), arr, len(steps), float(eta),
                                                 _ffi.ptr(x_T_noise), _ffi.ptr(step_noise), int(seed), int(shape_offset), ri, _ffi.ptr(traj),
                                                 _ffi.ptr(pred), B, N, _ffi.current_stream())
        _ffi.check(rc, "dfx_sample_chain_ddim")
        return pred, traj

    def snapshot_times(self, ret_interval):
        T = self.num_timesteps
        nk = T // ret_interval
        return [(nk - k) * ret_interval for k in range(nk)]

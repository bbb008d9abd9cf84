"""Host-side driver of libdfx's latent sampler: owns the opaque ``dfx_latents`` handle.

    ls = LatentSampler(params, n_class=4, zdim=256, n_heads=8, d_head=32, noise_scale=100.0)
    out = ls.sample_latents(w_noise, aligner_noise, valid_id, fixed_id, K=10, npoints=2048)

``params`` maps the reference ``state_dict`` names relative to ``encoder.`` (``flow.{i}.chain.{l}.net_s_t.{0,2,4}.*``,
``part_aligner.*``; python/difffacto/models/encoders/flow.py:9-19, part_encoders.py:52-86) to fp32 tensors.
PyTorch is used for device memory and the stream only; the random draws are inputs.
"""
import ctypes

import numpy as np
import torch

from . import _ffi

_BLOCK_FIELDS = {
    "norm2_w": "norm2.weight", "norm2_b": "norm2.bias", "to_q": "attn2.to_q.weight", "to_k": "attn2.to_k.weight",
    "to_v": "attn2.to_v.weight", "to_out_w": "attn2.to_out.0.weight", "to_out_b": "attn2.to_out.0.bias",
    "norm3_w": "norm3.weight", "norm3_b": "norm3.bias", "ff_proj_w": "ff.net.0.proj.weight",
    "ff_proj_b": "ff.net.0.proj.bias", "ff_out_w": "ff.net.2.weight", "ff_out_b": "ff.net.2.bias",
}
_TOP_FIELDS = {
    "proj_in_w": "proj_in.weight", "proj_in_b": "proj_in.bias", "class_emb": "class_emb.weight",
    "pre_norm_w": "pre_norm.weight", "pre_norm_b": "pre_norm.bias", "post_norm_w": "post_norm.weight",
    "post_norm_b": "post_norm.bias", "proj_out_w": "proj_out.weight", "proj_out_b": "proj_out.bias",
}
_FLOW_KEYS = ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")


class LatentSampler:
    def __init__(self, params, n_class=4, zdim=256, n_heads=8, d_head=32, cimle=True, noise_dim=32,
                 noise_scale=10.0, prior_var=1.0, log_scale_var=0.0, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("LatentSampler needs a HIP device (there is no CPU path)")
        self.device = torch.device(device if device is not None else "cuda")
        self.n_class, self.zdim, self.noise_dim, self.cimle = int(n_class), int(zdim), int(noise_dim), bool(cimle)
        inner = n_heads * d_head
        flow_depth = 0
        while f"flow.0.chain.{flow_depth}.net_s_t.0.weight" in params:
            flow_depth += 1
        depth = 0
        while f"part_aligner.transformer_blocks.{depth}.norm2.weight" in params:
            depth += 1
        if not 1 <= depth <= _ffi.DFX_MAX_DEPTH:
            raise RuntimeError(f"unsupported aligner depth {depth}")
        keep = []

        def dev(key, shape=None):
            t = params[key]
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(np.asarray(t))
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            if shape is not None and tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"{key}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
            keep.append(t)
            return t.data_ptr()

        w = _ffi.LatentWeights()
        w.n_class, w.zdim, w.flow_depth = self.n_class, self.zdim, flow_depth
        half = zdim // 2
        flow_ptrs = None
        hidden = 0
        if flow_depth:
            hidden = int(params["flow.0.chain.0.net_s_t.0.weight"].shape[0])
            shapes = ((hidden, half), (hidden,), (hidden, hidden), (hidden,), (2 * half, hidden), (2 * half,))
            flow_ptrs = (_ffi.c_fp * (self.n_class * flow_depth * 6))()
            for p in range(self.n_class):
                for l in range(flow_depth):
                    for k, (suffix, shp) in enumerate(zip(_FLOW_KEYS, shapes)):
                        flow_ptrs[(p * flow_depth + l) * 6 + k] = dev(f"flow.{p}.chain.{l}.net_s_t.{suffix}", shp)
            w.flow = ctypes.cast(flow_ptrs, ctypes.POINTER(_ffi.c_fp))
        w.flow_hidden = hidden
        w.depth, w.n_heads, w.d_head = depth, int(n_heads), int(d_head)
        w.cimle, w.noise_dim = int(self.cimle), self.noise_dim
        w.noise_scale, w.prior_var, w.log_scale_var = float(noise_scale), float(prior_var), float(log_scale_var)
        in_ch = zdim + (noise_dim if cimle else 0)
        top_shapes = {"proj_in_w": (inner, in_ch), "class_emb": (self.n_class, inner), "proj_out_w": (6, inner)}
        for field, key in _TOP_FIELDS.items():
            setattr(w, field, dev("part_aligner." + key, top_shapes.get(field)))
        blk_shapes = {"to_q": (inner, inner), "to_k": (inner, inner), "to_v": (inner, inner), "to_out_w": (inner, inner),
                      "ff_proj_w": (8 * inner, inner), "ff_out_w": (inner, 4 * inner)}
        for b in range(depth):
            for field, key in _BLOCK_FIELDS.items():
                setattr(w.blocks[b], field, dev(f"part_aligner.transformer_blocks.{b}.{key}", blk_shapes.get(field)))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_latents_create(ctypes.byref(handle), ctypes.byref(w), _ffi.current_stream())
        _ffi.check(rc, "dfx_latents_create")
        del keep, flow_ptrs   # create() synchronised the stream
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().dfx_latents_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _f(self, t):
        return None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def flow_reverse(self, w):
        """w (S,zdim,n_class) standard normal -> part_code (S,zdim,n_class) (part_encoders.py:1054-1060)."""
        w = self._f(w)
        S = w.shape[0]
        assert tuple(w.shape) == (S, self.zdim, self.n_class)
        out = torch.empty_like(w)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_flow_reverse(self._h, _ffi.ptr(w), _ffi.ptr(out), S, _ffi.current_stream())
        _ffi.check(rc, "dfx_flow_reverse")
        return out

    def part_aligner(self, part_code, valid_id, noise=None):
        """PartAlignerTransformer.forward (part_encoders.py:88-109) -> mean (B,3,J), logvar (B,3,J)."""
        part_code, valid_id, noise = self._f(part_code), self._f(valid_id), self._f(noise)
        B = part_code.shape[0]
        assert tuple(part_code.shape) == (B, self.zdim, self.n_class) and tuple(valid_id.shape) == (B, self.n_class)
        if noise is not None:
            assert tuple(noise.shape) == (B, self.noise_dim)
        mean = torch.empty(B, 3, self.n_class, dtype=torch.float32, device=self.device)
        logvar = torch.empty_like(mean)
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_part_aligner(self._h, _ffi.ptr(part_code), _ffi.ptr(valid_id), _ffi.ptr(noise),
                                             _ffi.ptr(mean), _ffi.ptr(logvar), B, _ffi.current_stream())
        _ffi.check(rc, "dfx_part_aligner")
        return mean, logvar

    def sample_latents(self, w_noise, aligner_noise, valid_id, fixed_id=None, K=1, npoints=2048, part_code=None):
        """Everything of sample_latents after the random draws (part_encoders.py:1052-1110); dict of tensors with
        R = S*K rows: part_code, valid_id, noise, mean, logvar, params (= ctx[1]), seg_mask, mean_per_point,
        logvar_per_point."""
        w_noise, part_code, aligner_noise, valid_id = map(self._f, (w_noise, part_code, aligner_noise, valid_id))
        S = valid_id.shape[0]
        J, Z, R = self.n_class, self.zdim, S * int(K)
        for t in (w_noise, part_code):
            assert t is None or tuple(t.shape) == (S, Z, J)
        if aligner_noise is not None:
            assert tuple(aligner_noise.shape) == (R, self.noise_dim)
        fid = (ctypes.c_int32 * J)(*([0] * J if fixed_id is None else [int(v) for v in fixed_id]))
        e = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device=self.device)
        out = {"part_code": e(R, Z, J), "valid_id": e(R, J), "noise": e(R, self.noise_dim) if self.cimle else None,
               "mean": e(R, 3, J), "logvar": e(R, 3, J), "params": e(R, 6, J), "seg_mask": e(R, npoints, dtype=torch.int32),
               "mean_per_point": e(R, 3, npoints), "logvar_per_point": e(R, 3, npoints)}
        with torch.cuda.device(self.device):
            rc = _ffi.lib().dfx_sample_latents(
                self._h, _ffi.ptr(w_noise), _ffi.ptr(part_code), _ffi.ptr(aligner_noise), _ffi.ptr(valid_id), fid, S,
                int(K), int(npoints), _ffi.ptr(out["part_code"]), _ffi.ptr(out["valid_id"]), _ffi.ptr(out["noise"]),
                _ffi.ptr(out["mean"]), _ffi.ptr(out["logvar"]), _ffi.ptr(out["params"]), _ffi.ptr(out["seg_mask"]),
                _ffi.ptr(out["mean_per_point"]), _ffi.ptr(out["logvar_per_point"]), _ffi.current_stream())
        _ffi.check(rc, "dfx_sample_latents")
        return out

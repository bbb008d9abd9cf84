"""PointNet++ set-abstraction / feature-propagation modules with the reference's constructor
signatures and ``state_dict`` layout (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-209):
``mlps.{k}.{0,1,2,...}`` = Conv2d(1x1, bias = not bn) / BatchNorm2d / ReLU triples, ``groupers.{k}``.

Sampling / neighbourhood search / gathers run on libdfx's gfx950 kernels through ``pointnet2_utils``.
In inference (``module.eval()`` under ``torch.no_grad()``) the grouping, the shared per-neighbour MLP
(Conv2d 1x1 + BatchNorm2d with running statistics + ReLU) and the max over the neighbourhood run in
libdfx as well (``dfx_sa_forward_f32`` / ``dfx_fp_forward_f32``: one fused launch for the SA1 / SA2
shapes of PointNet2SSG); the module's torch parameters are only read (so checkpoints load unchanged).
When gradients or batch statistics are needed (``train()`` or grad mode) the MLP runs through the
module's own torch layers on the device, as in the reference.
"""
import ctypes
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _ffi
from . import pointnet2_utils as pu


def build_shared_mlp(mlp_spec: List[int], bn: bool = True) -> nn.Sequential:
    """1x1 Conv2d (+ BatchNorm2d) + ReLU per consecutive channel pair (pointnet2_modules.py:9-19)."""
    seq = nn.Sequential()
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        seq.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            seq.append(nn.BatchNorm2d(cout))
        seq.append(nn.ReLU(True))
    return seq


class _NativeMLP:
    """libdfx handle (dfx_shared_mlp) for a ``build_shared_mlp`` stack, rebuilt when a parameter / buffer changes."""

    def __init__(self, seq: nn.Sequential):
        self.layers = []   # (conv, bn or None)
        mods = list(seq)
        i = 0
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, nn.Conv2d) or conv.kernel_size != (1, 1):
                raise NotImplementedError("native shared MLP: expected Conv2d 1x1")
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
            self.layers.append((conv, bn))
            i += 3 if bn is not None else 2
        self._h = None
        self._ver = None

    def _tensors(self):
        out = []
        for conv, bn in self.layers:
            out += [conv.weight, conv.bias]
            if bn is not None:
                out += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return [t for t in out if t is not None]

    def handle(self):
        ts = self._tensors()
        ver = tuple((t._version, t.data_ptr()) for t in ts)
        if self._h is None or ver != self._ver:
            self.close()
            L = len(self.layers)
            ch = (ctypes.c_int32 * (L + 1))(self.layers[0][0].in_channels, *[c.out_channels for c, _ in self.layers])
            keep = []

            def arr(get):
                a = (_ffi.c_fp * L)()
                for l, (conv, bn) in enumerate(self.layers):
                    t = get(conv, bn)
                    if t is not None:
                        t = t.detach().to(torch.float32).contiguous()
                        keep.append(t)
                        a[l] = t.data_ptr()
                return a

            args = [arr(lambda c, b: c.weight.reshape(c.out_channels, c.in_channels)), arr(lambda c, b: c.bias),
                    arr(lambda c, b: None if b is None else b.weight), arr(lambda c, b: None if b is None else b.bias),
                    arr(lambda c, b: None if b is None else b.running_mean), arr(lambda c, b: None if b is None else b.running_var)]
            eps = next((b.eps for _, b in self.layers if b is not None), 1e-5)
            h = ctypes.c_void_p()
            dev = self.layers[0][0].weight.device
            with torch.cuda.device(dev):
                rc = _ffi.lib().dfx_shared_mlp_create(ctypes.byref(h), L, ch, *[ctypes.cast(a, ctypes.POINTER(_ffi.c_fp)) for a in args],
                                                      float(eps), 0xFFFFFFFF, _ffi.current_stream())
            _ffi.check(rc, "dfx_shared_mlp_create")
            self._h, self._ver = h, ver
        return self._h

    def close(self):
        if self._h is not None:
            _ffi.lib().dfx_shared_mlp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _inference(module: nn.Module) -> bool:
    return not module.training and not torch.is_grad_enabled()


class _PointnetSAModuleBase(nn.Module):
    """FPS -> gather centres -> per scale: group -> shared MLP -> max over the neighbourhood
    (pointnet2_modules.py:22-74)."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _centres(self, xyz: torch.Tensor) -> Optional[torch.Tensor]:
        if self.npoint is None:
            return None
        picked = pu.furthest_point_sample(xyz, self.npoint)                       # (B, npoint) int32
        centres = pu.gather_operation(xyz.transpose(1, 2).contiguous(), picked)  # (B, 3, npoint)
        return centres.transpose(1, 2).contiguous()

    def _native(self, k: int) -> _NativeMLP:
        cache = self.__dict__.setdefault("_native_mlps", {})
        if k not in cache:
            cache[k] = _NativeMLP(self.mlps[k])
        return cache[k]

    def _forward_native(self, k, xyz, new_xyz, features, force_general=False):
        """grouper -> mlp -> max over nsample in libdfx (eval-mode BatchNorm)."""
        grouper = self.groupers[k]
        B, N, _ = xyz.shape
        pu._chk(xyz, "xyz", torch.float32)
        if features is not None:
            pu._chk(features, "features", torch.float32)
        C = 0 if features is None else features.shape[1]
        if isinstance(grouper, pu.QueryAndGroup):
            idx = pu.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
            M, ns, centres = new_xyz.shape[1], grouper.nsample, new_xyz
        else:
            idx, M, ns, centres = None, 1, N, None
        mlp = self._native(k)
        cout = mlp.layers[-1][0].out_channels
        out = torch.empty(B, cout, M, dtype=torch.float32, device=xyz.device)
        with torch.cuda.device(xyz.device):
            rc = _ffi.lib().dfx_sa_forward_f32(mlp.handle(), _ffi.ptr(xyz), _ffi.ptr(centres), _ffi.ptr(features), _ffi.ptr(idx),
                                               int(grouper.use_xyz), _ffi.ptr(out), B, N, M, ns, C, int(force_general),
                                               _ffi.current_stream())
        _ffi.check(rc, "dfx_sa_forward_f32")
        return out

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        new_xyz = self._centres(xyz)
        pooled = []
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if _inference(self):
                pooled.append(self._forward_native(k, xyz, new_xyz, features))
            else:
                nbh = mlp(grouper(xyz, new_xyz, features))   # (B, mlp[-1], npoint, nsample)
                pooled.append(nbh.amax(dim=3))               # max_pool2d over nsample, squeezed
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping SA layer: ``(npoint, radii, nsamples, mlps, bn=True, use_xyz=True)``
    (pointnet2_modules.py:77-115)."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError("radii, nsamples and mlps must have the same length")
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pu.QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None
                                 else pu.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3   # in place, like the reference (callers observe the mutated spec)
            self.mlps.append(build_shared_mlp(spec, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA layer: ``(mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True)``
    (pointnet2_modules.py:118-146)."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the 3 nearest known points, concat
    skip features, shared MLP (pointnet2_modules.py:149-209)."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = build_shared_mlp(mlp, bn=bn)

    def _forward_native(self, unknown, known, unknow_feats, known_feats):
        if "_native_mlp" not in self.__dict__:
            self.__dict__["_native_mlp"] = _NativeMLP(self.mlp)
        mlp = self.__dict__["_native_mlp"]
        B, n, _ = unknown.shape
        for t, name in ((unknown, "unknown"), (known, "known"), (unknow_feats, "unknow_feats"), (known_feats, "known_feats")):
            if t is not None:
                pu._chk(t, name, torch.float32)
        m = 1 if known is None else known.shape[1]
        C1 = 0 if unknow_feats is None else unknow_feats.shape[1]
        C2 = known_feats.shape[1]
        out = torch.empty(B, mlp.layers[-1][0].out_channels, n, dtype=torch.float32, device=unknown.device)
        with torch.cuda.device(unknown.device):
            rc = _ffi.lib().dfx_fp_forward_f32(mlp.handle(), _ffi.ptr(unknown), _ffi.ptr(known), _ffi.ptr(unknow_feats),
                                               _ffi.ptr(known_feats), _ffi.ptr(out), B, n, m, C1, C2, _ffi.current_stream())
        _ffi.check(rc, "dfx_fp_forward_f32")
        return out

    def forward(self, unknown, known, unknow_feats, known_feats):
        if _inference(self):
            return self._forward_native(unknown, known, unknow_feats, known_feats)
        if known is None:
            interp = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        else:
            dist, idx = pu.three_nn(unknown, known)
            inv = 1.0 / (dist + 1e-8)
            weight = inv / inv.sum(dim=2, keepdim=True)
            interp = pu.three_interpolate(known_feats, idx, weight)
        stacked = interp if unknow_feats is None else torch.cat([interp, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)

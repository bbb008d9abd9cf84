"""PointNet++ set-abstraction / feature-propagation modules with the reference's constructor
signatures and ``state_dict`` layout (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-209):
``mlps.{k}.{0,1,2,...}`` = Conv2d(1x1, bias = not bn) / BatchNorm2d / ReLU triples, ``groupers.{k}``.

Sampling / neighbourhood search / gathers run on libdfx's gfx950 kernels through ``pointnet2_utils``.
In inference (``module.eval()`` under ``torch.no_grad()``) the grouping, the shared per-neighbour MLP
(Conv2d 1x1 + BatchNorm2d with running statistics + ReLU) and the max over the neighbourhood run in
libdfx as well (``dfx_sa_forward_f32`` / ``dfx_fp_forward_f32``: one fused launch for the SA1 / SA2
shapes of PointNet2SSG); the module's torch parameters are only read (so checkpoints load unchanged).
In ``train()`` mode (round 6) the shared MLP with BATCH-statistics BatchNorm, its running-statistics update, the max
and their gradients run in libdfx too (``dfx_shared_mlp_train_forward`` / ``_backward`` behind an autograd Function;
the grouping / interpolation already had native gradient kernels); ``eval()`` with gradients enabled takes the same kernels
with the running statistics.  Only stacks the training kernels do not serve (an output width that is not a multiple of 4)
run the module's own torch layers.
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


def _train_native_ok(seq: nn.Sequential) -> bool:
    """The training kernels (dfx_shared_mlp_train_*) serve build_shared_mlp stacks of <= 4 layers whose output widths are multiples of 4, fp32 on the
    GPU, BatchNorm in train() mode (batch statistics) or in eval() mode under autograd (running statistics); anything else stays on the torch layers."""
    try:
        layers = _NativeMLP(seq).layers
    except NotImplementedError:
        return False
    if not 1 <= len(layers) <= _ffi.DFX_MLP_MAX_LAYERS:
        return False
    for conv, bn in layers:
        if conv.out_channels % 4 or conv.out_channels > 1024 or conv.weight.dtype != torch.float32 or not conv.weight.is_cuda:
            return False
        if bn is not None and (not bn.affine or not bn.track_running_stats or bn.momentum is None):
            return False
    modes = {bn.training for _, bn in layers if bn is not None}
    return len(modes) <= 1   # (all BatchNorm layers in the same mode: batch statistics or running statistics)


class _SharedMLPTrainFn(torch.autograd.Function):
    """y = [max over nsample of] mlp(x) for a stack of (1x1 convolution / linear layer [+ BatchNorm] [+ ReLU]) on libdfx (dfx_shared_mlp_train_forward /
    _backward): what autograd through nn.Conv2d(1x1) / nn.BatchNorm2d / ReLU / F.max_pool2d computes in the reference (pointnet2_modules.py:9-19, :62-70).
    x (B, C, M, ns) -> (B, C_out, M) if pool else (B, C_out, M, ns).  ``spec``: per layer (c_out, c_in, has_bias, bn module or None); ``params``: the
    layers' weight [, bias] [, bn.weight, bn.bias] in order; BatchNorm on batch statistics when its module is in train() mode (running statistics
    updated in place), on the running statistics otherwise."""

    @staticmethod
    def forward(ctx, x, pool, spec, relu_mask, *params):
        pu._chk(x, "grouped features", torch.float32)
        B, C, M, ns = x.shape
        desc = _ffi.SharedMlpTrain()
        desc.layers = len(spec)
        desc.ch[0] = C
        desc.relu_mask = int(relu_mask)
        keep, k, slots = [], 0, []
        for l, (cout, cin, has_bias, bn) in enumerate(spec):
            desc.ch[l + 1] = cout
            w = params[k].detach().reshape(cout, cin).contiguous()
            slot = {"w": (tuple(params[k].shape), w.numel())}
            k += 1
            keep.append(w)
            desc.conv_w[l] = w.data_ptr()
            if has_bias:
                b = params[k].detach().contiguous()
                k += 1
                keep.append(b)
                desc.conv_b[l] = b.data_ptr()
                slot["b"] = cout
            if bn is not None:
                g, be = params[k].detach().contiguous(), params[k + 1].detach().contiguous()
                k += 2
                keep += [g, be]
                desc.bn_w[l], desc.bn_b[l] = g.data_ptr(), be.data_ptr()
                desc.bn_mean[l], desc.bn_var[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                desc.bn_eps = float(bn.eps)
                slot["bn"] = cout
            slots.append(slot)
        bns = [bn for _, _, _, bn in spec if bn is not None]
        momentum = float(bns[0].momentum) if bns else -1.0
        batch_stats = int(bns[0].training) if bns else 1
        lib = _ffi.lib()
        nbytes = lib.dfx_shared_mlp_train_workspace_bytes(ctypes.byref(desc), B, M, ns)
        if nbytes == 0:
            raise RuntimeError("dfx_shared_mlp_train_workspace_bytes: unsupported configuration")
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
        wsp = (ws.data_ptr() + 255) & ~255
        cout = spec[-1][0]
        out = torch.empty((B, cout, M) if pool else (B, cout, M, ns), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dfx_shared_mlp_train_forward(ctypes.byref(desc), ctypes.c_void_p(wsp), nbytes, _ffi.ptr(x), _ffi.ptr(out), B, M, ns, int(bool(pool)),
                                                  batch_stats, momentum, _ffi.current_stream())
        _ffi.check(rc, "dfx_shared_mlp_train_forward")
        for bn in bns:
            if batch_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        ctx.desc, ctx.keep, ctx.ws, ctx.wsp, ctx.nbytes, ctx.slots, ctx.dims, ctx.pool, ctx.batch_stats = desc, keep, ws, wsp, nbytes, slots, (B, C, M, ns), bool(pool), batch_stats
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, C, M, ns = ctx.dims
        d_out = d_out.contiguous()
        dev = d_out.device
        g = _ffi.SharedMlpTrain()
        g.layers = ctx.desc.layers
        grads = []
        for l, slot in enumerate(ctx.slots):
            shape, n = slot["w"]
            dw = torch.empty(n, dtype=torch.float32, device=dev)
            g.conv_w[l] = dw.data_ptr()
            grads.append(dw.view(shape))
            if "b" in slot:
                db = torch.empty(slot["b"], dtype=torch.float32, device=dev)
                g.conv_b[l] = db.data_ptr()
                grads.append(db)
            if "bn" in slot:
                dg, dbe = torch.empty(slot["bn"], dtype=torch.float32, device=dev), torch.empty(slot["bn"], dtype=torch.float32, device=dev)
                g.bn_w[l], g.bn_b[l] = dg.data_ptr(), dbe.data_ptr()
                grads += [dg, dbe]
        d_x = torch.empty(B, C, M, ns, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(dev):
            rc = _ffi.lib().dfx_shared_mlp_train_backward(ctypes.byref(ctx.desc), ctypes.c_void_p(ctx.wsp), ctx.nbytes, _ffi.ptr(d_out), ctypes.byref(g), _ffi.ptr(d_x),
                                                          B, M, ns, int(ctx.pool), ctx.batch_stats, _ffi.current_stream())
        _ffi.check(rc, "dfx_shared_mlp_train_backward")
        return (d_x, None, None, None, *grads)


def mlp_train(stack, x: torch.Tensor, pool: bool, relu_mask: Optional[int] = None) -> torch.Tensor:
    """``stack``: [(layer, bn or None)] with ``layer`` an nn.Conv2d(1x1) or nn.Linear; runs it on libdfx under autograd (``_SharedMLPTrainFn``)."""
    spec, params = [], []
    for layer, bn in stack:
        cout, cin = (layer.out_features, layer.in_features) if isinstance(layer, nn.Linear) else (layer.out_channels, layer.in_channels)
        spec.append((cout, cin, layer.bias is not None, bn))
        params.append(layer.weight)
        if layer.bias is not None:
            params.append(layer.bias)
        if bn is not None:
            params += [bn.weight, bn.bias]
    if relu_mask is None:
        relu_mask = (1 << len(spec)) - 1
    return _SharedMLPTrainFn.apply(x.contiguous(), pool, spec, relu_mask, *params)


def shared_mlp_train(seq: nn.Sequential, x: torch.Tensor, pool: bool) -> torch.Tensor:
    """``mlp(x)`` [+ max over the last axis] of a ``build_shared_mlp`` stack under autograd on libdfx (see ``_SharedMLPTrainFn``)."""
    return mlp_train(_NativeMLP(seq).layers, x, pool)


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
            elif xyz.is_cuda and _train_native_ok(mlp):
                # train() / eval() under autograd: grouping (libdfx, with its gradient kernels) -> Conv2d 1x1 + BatchNorm2d (batch statistics) + ReLU -> max, natively (round 6)
                pooled.append(shared_mlp_train(mlp, grouper(xyz, new_xyz, features), pool=True))
            else:   # a stack the training kernels do not serve (an output width that is not a multiple of 4, mixed BatchNorm modes)
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
        if stacked.is_cuda and _train_native_ok(self.mlp):
            return shared_mlp_train(self.mlp, stacked.unsqueeze(-1), pool=False).squeeze(-1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)

"""Autograd wrappers with the reference's names, signatures and error behaviour
(pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:34-379), calling libdfx through ctypes.

Conventions kept from the reference's C++ layer (``_ext-src/src/*.cpp``, ``include/utils.h:5-25``):
float32 / int32, contiguous, device tensors only; violations raise ``RuntimeError``; outputs are
freshly allocated; every launch goes on torch's current stream.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _ffi


def _chk(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: CPU not supported")          # AT_ASSERT(false, "CPU not supported")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")  # CHECK_CONTIGUOUS
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {'float' if dtype == torch.float32 else 'int'} tensor")  # CHECK_IS_*


def _run(name, *args):
    L = _ffi.lib()
    _ffi.check(getattr(L, name)(*args, _ffi.current_stream()), name)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        r"""xyz (B, N, 3) float32 -> (B, npoint) int32 indices (pointnet2_utils.py:34-65)."""
        _chk(xyz, "points", torch.float32)
        B, N, _ = xyz.shape
        out = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        tmp = None
        if N > _ffi.lib().dfx_fps_max_resident():
            tmp = torch.empty(B, N, dtype=torch.float32, device=xyz.device)
        _run("dfx_furthest_point_sampling_f32", _ffi.ptr(xyz), _ffi.ptr(tmp), _ffi.ptr(out), B, N, int(npoint))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        r"""features (B, C, N), idx (B, npoint) int32 -> (B, C, npoint) (pointnet2_utils.py:68-101)."""
        _chk(features, "points", torch.float32)
        _chk(idx, "idx", torch.int32)
        ctx.save_for_backward(idx, features)
        B, C, N = features.shape
        M = idx.shape[1]
        out = torch.empty(B, C, M, dtype=torch.float32, device=features.device)
        _run("dfx_gather_points_f32", _ffi.ptr(features), _ffi.ptr(idx), _ffi.ptr(out), B, C, N, M)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, C, N = features.shape
        grad_out = grad_out.contiguous()
        _chk(grad_out, "grad_out", torch.float32)
        g = torch.empty(B, C, N, dtype=torch.float32, device=grad_out.device)
        _run("dfx_gather_points_grad_f32", _ffi.ptr(grad_out), _ffi.ptr(idx), _ffi.ptr(g), B, C, N, idx.shape[1])
        return g, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        r"""unknown (B, n, 3), known (B, m, 3) -> (dist (B, n, 3) = sqrt(dist2), idx (B, n, 3))
        (pointnet2_utils.py:104-133)."""
        _chk(unknown, "unknowns", torch.float32)
        _chk(known, "knows", torch.float32)
        B, n, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknown.device)
        _run("dfx_three_nn_f32", _ffi.ptr(unknown), _ffi.ptr(known), _ffi.ptr(dist2), _ffi.ptr(idx), B, n, m)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        r"""features (B, c, m), idx (B, n, 3), weight (B, n, 3) -> (B, c, n) (pointnet2_utils.py:136-191)."""
        _chk(features, "points", torch.float32)
        _chk(idx, "idx", torch.int32)
        _chk(weight, "weight", torch.float32)
        ctx.save_for_backward(idx, weight, features)
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        _run("dfx_three_interpolate_f32", _ffi.ptr(features), _ffi.ptr(idx), _ffi.ptr(weight), _ffi.ptr(out), B, c, m, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, features = ctx.saved_tensors
        B, c, m = features.shape
        n = idx.shape[1]
        grad_out = grad_out.contiguous()
        _chk(grad_out, "grad_out", torch.float32)
        g = torch.empty(B, c, m, dtype=torch.float32, device=grad_out.device)
        _run("dfx_three_interpolate_grad_f32", _ffi.ptr(grad_out), _ffi.ptr(idx), _ffi.ptr(weight), _ffi.ptr(g), B, c, n, m)
        return g, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        r"""features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample) (pointnet2_utils.py:194-240)."""
        _chk(features, "points", torch.float32)
        _chk(idx, "idx", torch.int32)
        ctx.save_for_backward(idx, features)
        B, C, N = features.shape
        _, npoint, nsample = idx.shape
        out = torch.empty(B, C, npoint, nsample, dtype=torch.float32, device=features.device)
        _run("dfx_group_points_f32", _ffi.ptr(features), _ffi.ptr(idx), _ffi.ptr(out), B, C, N, npoint, nsample)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, C, N = features.shape
        _, npoint, nsample = idx.shape
        grad_out = grad_out.contiguous()
        _chk(grad_out, "grad_out", torch.float32)
        g = torch.empty(B, C, N, dtype=torch.float32, device=grad_out.device)
        _run("dfx_group_points_grad_f32", _ffi.ptr(grad_out), _ffi.ptr(idx), _ffi.ptr(g), B, C, N, npoint, nsample)
        return g, torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        r"""radius, nsample, xyz (B, N, 3), new_xyz (B, npoint, 3) -> (B, npoint, nsample) int32.
        NB the python argument order differs from the C++ op (pointnet2_utils.py:243-276)."""
        _chk(new_xyz, "new_xyz", torch.float32)
        _chk(xyz, "xyz", torch.float32)
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        out = torch.empty(B, M, int(nsample), dtype=torch.int32, device=xyz.device)
        _run("dfx_ball_query_f32", _ffi.ptr(new_xyz), _ffi.ptr(xyz), _ffi.ptr(out), B, N, M, float(radius), int(nsample))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


def _stack_neighbourhood(rel_xyz, feats, use_xyz):
    """Channel-concatenate [centre-relative xyz | features] the way both grouper modules do."""
    parts = []
    if feats is None or use_xyz:
        parts.append(rel_xyz)
    if feats is not None:
        parts.append(feats)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


class QueryAndGroup(nn.Module):
    r"""``QueryAndGroup(radius, nsample, use_xyz=True)`` of the reference (pointnet2_utils.py:279-333):
    ball query around ``new_xyz`` (B, npoint, 3), gather neighbour coordinates (made relative to the
    centre) and neighbour features -> (B, 3 + C, npoint, nsample)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius = radius
        self.nsample = nsample
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        if features is None and not self.use_xyz:
            raise AssertionError("Cannot have not features and not use xyz as a feature!")
        nbr = ball_query(self.radius, self.nsample, xyz, new_xyz)                    # (B, npoint, nsample)
        rel = grouping_operation(xyz.transpose(1, 2).contiguous(), nbr)              # (B, 3, npoint, nsample)
        rel = rel - new_xyz.transpose(1, 2).unsqueeze(-1)
        feats = grouping_operation(features, nbr) if features is not None else None
        return _stack_neighbourhood(rel, feats, self.use_xyz)


class GroupAll(nn.Module):
    r"""``GroupAll(use_xyz=True)`` (pointnet2_utils.py:336-379): the whole cloud is one neighbourhood,
    output (B, 3 + C, 1, N); ``new_xyz`` is ignored."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        whole = xyz.transpose(1, 2).unsqueeze(2)                                     # (B, 3, 1, N)
        feats = features.unsqueeze(2) if features is not None else None
        return _stack_neighbourhood(whole, feats, self.use_xyz)

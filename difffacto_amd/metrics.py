"""Chamfer-L2 on libdfx — mirrors ``ChamferFunction`` / ``ChamferDistanceL2`` / ``ChamferDistanceL2_split`` of
python/difffacto/metrics/chamfer_dist/__init__.py:14-73 (forward + backward through the HIP kernels)."""
import torch

from . import _ffi


def _chk(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: CPU not supported")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous {dtype} tensor")


class ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        _chk(xyz1, "xyz1"), _chk(xyz2, "xyz2")
        B, N, _ = xyz1.shape
        M = xyz2.shape[1]
        d1 = torch.empty(B, N, device=xyz1.device)
        d2 = torch.empty(B, M, device=xyz1.device)
        i1 = torch.empty(B, N, dtype=torch.int32, device=xyz1.device)
        i2 = torch.empty(B, M, dtype=torch.int32, device=xyz1.device)
        _ffi.check(_ffi.lib().dfx_chamfer_forward_f32(_ffi.ptr(xyz1), _ffi.ptr(xyz2), _ffi.ptr(d1), _ffi.ptr(d2),
                                                      _ffi.ptr(i1), _ffi.ptr(i2), B, N, M, _ffi.current_stream()),
                   "dfx_chamfer_forward_f32")
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        return d1, d2

    @staticmethod
    def backward(ctx, g1, g2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        g1, g2 = g1.contiguous(), g2.contiguous()
        B, N, _ = xyz1.shape
        M = xyz2.shape[1]
        gx1, gx2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        _ffi.check(_ffi.lib().dfx_chamfer_backward_f32(_ffi.ptr(xyz1), _ffi.ptr(xyz2), _ffi.ptr(i1), _ffi.ptr(i2),
                                                       _ffi.ptr(g1), _ffi.ptr(g2), _ffi.ptr(gx1), _ffi.ptr(gx2), B, N, M,
                                                       _ffi.current_stream()), "dfx_chamfer_backward_f32")
        return gx1, gx2


class ChamferDistanceL2(torch.nn.Module):
    """mean(dist1) + mean(dist2) (``reduce=True``) — chamfer_dist/__init__.py:29-52."""

    def __init__(self, ignore_zeros=False, reduce=True):
        super().__init__()
        self.ignore_zeros, self.reduce = ignore_zeros, reduce

    def _dists(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(0)
            xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(0)
        return ChamferFunction.apply(xyz1, xyz2)

    def forward(self, xyz1, xyz2):
        d1, d2 = self._dists(xyz1, xyz2)
        if self.reduce:
            d1, d2 = d1.mean(), d2.mean()
        return d1 + d2


class ChamferDistanceL2_split(ChamferDistanceL2):
    """Returns (dist1, dist2) separately — chamfer_dist/__init__.py:54-73."""

    def forward(self, xyz1, xyz2):
        d1, d2 = self._dists(xyz1, xyz2)
        if self.reduce:
            d1, d2 = d1.mean(), d2.mean()
        return d1, d2


def chamfer_l2(a, b):
    """Per-cloud symmetric Chamfer-L2 (B,) between (B,N,3) and (B,M,3): mean_i d1 + mean_j d2."""
    d1, d2 = ChamferFunction.apply(a, b)
    return d1.mean(dim=1) + d2.mean(dim=1)

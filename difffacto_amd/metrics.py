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


# ---------------------------------------------------------------------------------------------------------------------
# Approximate EMD by auction — mirrors ``emdFunction`` / ``EMD`` of python/difffacto/metrics/emd/emd_module.py:17-70
class emdFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters):
        """xyz1, xyz2 (B,n,3) in [0,1]^3 -> (dist (B,n) squared matched distances, assignment (B,n) int32)."""
        assert xyz1.shape == xyz2.shape and xyz1.shape[-1] == 3
        xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
        _chk(xyz1, "xyz1"), _chk(xyz2, "xyz2")
        B, n, _ = xyz1.shape
        dist = torch.empty(B, n, device=xyz1.device)
        assignment = torch.empty(B, n, dtype=torch.int32, device=xyz1.device)
        ws = torch.empty(max(_ffi.lib().dfx_emd_workspace_bytes(B, n), 4), dtype=torch.uint8, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            _ffi.check(_ffi.lib().dfx_emd_forward_f32(_ffi.ptr(xyz1), _ffi.ptr(xyz2), _ffi.ptr(dist), _ffi.ptr(assignment), _ffi.ptr(ws),
                                                      B, n, float(eps), int(iters), _ffi.current_stream()), "dfx_emd_forward_f32")
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        graddist = graddist.contiguous()
        B, n, _ = xyz1.shape
        g1 = torch.empty_like(xyz1)
        with torch.cuda.device(xyz1.device):
            _ffi.check(_ffi.lib().dfx_emd_backward_f32(_ffi.ptr(xyz1), _ffi.ptr(xyz2), _ffi.ptr(graddist), _ffi.ptr(assignment),
                                                       _ffi.ptr(g1), B, n, _ffi.current_stream()), "dfx_emd_backward_f32")
        return g1, torch.zeros_like(xyz2), None, None   # the reference leaves grad_xyz2 at zero (emd_module.py:46-50)


class EMD(torch.nn.Module):
    """``METRICS['EMD']``: ``dist_only`` -> sqrt(dist).mean(1) per cloud pair (evaluation_utils.py:84-89 uses
    EMD(0.002, 10000, True)); otherwise (dist, assignment)."""

    def __init__(self, eps, iters, dist_only=False):
        super().__init__()
        self.eps, self.iters, self.dist_only = eps, iters, dist_only

    def forward(self, input1, input2):
        out = emdFunction.apply(input1, input2, self.eps, self.iters)
        return torch.sqrt(out[0]).mean(1) if self.dist_only else out

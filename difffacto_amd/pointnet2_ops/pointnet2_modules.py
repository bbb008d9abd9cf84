"""PointNet++ set-abstraction / feature-propagation modules with the reference's constructor
signatures and ``state_dict`` layout (pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-209):
``mlps.{k}.{0,1,2,...}`` = Conv2d(1x1, bias = not bn) / BatchNorm2d / ReLU triples, ``groupers.{k}``.

Sampling / neighbourhood search / gathers run on libdfx's gfx950 kernels through
``pointnet2_utils``; the shared per-neighbour MLP and the max over the neighbourhood use the
module's own torch parameters (so checkpoints load unchanged).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

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

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        new_xyz = self._centres(xyz)
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
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

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is None:
            interp = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        else:
            dist, idx = pu.three_nn(unknown, known)
            inv = 1.0 / (dist + 1e-8)
            weight = inv / inv.sum(dim=2, keepdim=True)
            interp = pu.three_interpolate(known_feats, idx, weight)
        stacked = interp if unknow_feats is None else torch.cat([interp, unknow_feats], dim=1)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)

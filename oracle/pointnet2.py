"""ctypes wrapper around the C oracle of the pointnet2_ops kernels (ORACLE — test only).

Allocation / fill conventions follow the reference's C++ wrappers:
  gather / group / interpolate outputs  torch::zeros      (sampling.cpp:25-27, group_points.cpp, interpolate.cpp)
  FPS scratch                           full(1e10)        (sampling.cpp:74-76)
  ball_query idx                        zeros             (ball_query.cpp:19-21)
The python-level ``three_nn`` returns sqrt(dist2) (pointnet2_utils.py:124-125).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_pn2.so")
_lib = None

_F = np.float32
_I = np.int32


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pointnet2.c")):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def opt_n_threads(n):
    return lib().oracle_opt_n_threads(int(n))


def gather_points(points, idx):
    points, idx = _c(points, _F), _c(idx, _I)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.zeros((B, C, M), _F)
    lib().oracle_gather_points(B, C, N, M, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, N):
    grad_out, idx = _c(grad_out, _F), _c(idx, _I)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, N), _F)
    lib().oracle_gather_points_grad(B, C, N, M, _p(grad_out), _p(idx), _p(out))
    return out


def furthest_point_sampling(xyz, npoint):
    xyz = _c(xyz, _F)
    B, N, _ = xyz.shape
    tmp = np.full((B, N), 1e10, _F)
    out = np.zeros((B, npoint), _I)
    lib().oracle_furthest_point_sampling(B, N, int(npoint), _p(xyz), _p(tmp), _p(out))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """python-level argument order (pointnet2_utils.py:265)."""
    xyz, new_xyz = _c(xyz, _F), _c(new_xyz, _F)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), _I)
    lib().oracle_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    points, idx = _c(points, _F), _c(idx, _I)
    B, C, N = points.shape
    _, NP, NS = idx.shape
    out = np.zeros((B, C, NP, NS), _F)
    lib().oracle_group_points(B, C, N, NP, NS, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, N):
    grad_out, idx = _c(grad_out, _F), _c(idx, _I)
    B, C, NP, NS = grad_out.shape
    out = np.zeros((B, C, N), _F)
    lib().oracle_group_points_grad(B, C, N, NP, NS, _p(grad_out), _p(idx), _p(out))
    return out


def three_nn(unknown, known):
    """ThreeNN.forward (pointnet2_utils.py:104-128): (sqrt(dist2), idx)."""
    d2, idx = three_nn_dist2(unknown, known)
    return np.sqrt(d2), idx


def three_nn_dist2(unknown, known):
    """What ``_ext.three_nn`` returns (interpolate_gpu.cu:13-69): squared distances."""
    unknown, known = _c(unknown, _F), _c(known, _F)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((B, n, 3), _F)
    idx = np.zeros((B, n, 3), _I)
    lib().oracle_three_nn(B, n, m, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _c(points, _F), _c(idx, _I), _c(weight, _F)
    B, C, m = points.shape
    n = idx.shape[1]
    out = np.zeros((B, C, n), _F)
    lib().oracle_three_interpolate(B, C, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _c(grad_out, _F), _c(idx, _I), _c(weight, _F)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), _F)
    lib().oracle_three_interpolate_grad(B, C, n, m, _p(grad_out), _p(idx), _p(weight), _p(out))
    return out


def chamfer_forward(xyz1, xyz2):
    """dist1 (B,N), dist2 (B,M), idx1, idx2 like chamfer.forward (metrics/chamfer_dist/__init__.py:16)."""
    xyz1, xyz2 = _c(xyz1, _F), _c(xyz2, _F)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    d1, d2 = np.zeros((B, N), _F), np.zeros((B, M), _F)
    i1, i2 = np.zeros((B, N), _I), np.zeros((B, M), _I)
    lib().oracle_chamfer_nn(B, N, M, _p(xyz1), _p(xyz2), _p(d1), _p(i1))
    lib().oracle_chamfer_nn(B, M, N, _p(xyz2), _p(xyz1), _p(d2), _p(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, idx1, idx2, g1, g2):
    xyz1, xyz2, g1, g2 = _c(xyz1, _F), _c(xyz2, _F), _c(g1, _F), _c(g2, _F)
    idx1, idx2 = _c(idx1, _I), _c(idx2, _I)
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    gx1, gx2 = np.zeros_like(xyz1), np.zeros_like(xyz2)
    lib().oracle_chamfer_grad(B, N, M, _p(xyz1), _p(xyz2), _p(g1), _p(idx1), _p(gx1), _p(gx2))
    lib().oracle_chamfer_grad(B, M, N, _p(xyz2), _p(xyz1), _p(g2), _p(idx2), _p(gx2), _p(gx1))
    return gx1, gx2


def emd_forward(xyz1, xyz2, eps, iters):
    """emdFunction.forward (metrics/emd/emd_module.py:17-41): (dist (B,n) squared distances to the matched point, assignment (B,n))."""
    xyz1, xyz2 = _c(xyz1, _F), _c(xyz2, _F)
    B, n, _ = xyz1.shape
    assert xyz2.shape == xyz1.shape
    dist = np.zeros((B, n), _F)
    assignment = np.zeros((B, n), _I)
    lib().oracle_emd_forward(B, n, _p(xyz1), _p(xyz2), ctypes.c_float(eps), int(iters), _p(dist), _p(assignment))
    return dist, assignment


def emd_unassigned_per_iteration(xyz1, xyz2, eps, iters):
    """How many points bid in each iteration of the first pair's auction (0 once it has converged) — lets a test check that it stops
    the auction inside the tail."""
    xyz1, xyz2 = _c(xyz1, _F), _c(xyz2, _F)
    B, n, _ = xyz1.shape
    dist, assignment, hist = np.zeros((B, n), _F), np.zeros((B, n), _I), np.zeros(iters, _I)
    lib().oracle_emd_forward_trace(B, n, _p(xyz1), _p(xyz2), ctypes.c_float(eps), int(iters), _p(dist), _p(assignment), _p(hist))
    return hist


def emd_backward(xyz1, xyz2, grad_dist, assignment):
    xyz1, xyz2, grad_dist, assignment = _c(xyz1, _F), _c(xyz2, _F), _c(grad_dist, _F), _c(assignment, _I)
    B, n, _ = xyz1.shape
    g = np.zeros_like(xyz1)
    lib().oracle_emd_backward(B, n, _p(xyz1), _p(xyz2), _p(grad_dist), _p(assignment), _p(g))
    return g

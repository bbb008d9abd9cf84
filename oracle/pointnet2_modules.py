"""numpy fp32 restatement of the PointNet++ set-abstraction / feature-propagation modules (ORACLE — test only).

Follows the reference's Python composition on top of the C oracle's primitive ops (oracle/pointnet2.c):

* ``build_shared_mlp``           pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:9-19
                                 (Conv2d 1x1 [bias = not bn] -> BatchNorm2d (eval: running stats, eps 1e-5) -> ReLU)
* ``QueryAndGroup.forward``      pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:296-333 ([xyz - centre | features])
* ``GroupAll.forward``           pointnet2_utils.py:349-381 (xyz NOT centred)
* ``_PointnetSAModuleBase.forward``  pointnet2_modules.py:29-74 (FPS -> gather -> group -> MLP -> max over nsample)
* ``PointnetFPModule.forward``   pointnet2_modules.py:170-209 (three_nn, w = (1/(d+1e-8)) / sum, interpolate, cat, MLP)

``W`` maps the module's ``state_dict`` names (``mlps.0.0.weight``, ``mlps.0.1.running_mean`` ...; FP: ``mlp.0.weight``)
to numpy arrays.  Pinned by tests/golden/sa_*.npz / fp_*.npz (tests/golden/make_golden_sa.py).
"""
import numpy as np

from . import pointnet2 as o

F32 = np.float32
BN_EPS = 1e-5


def shared_mlp(x, W, prefix, n_layers, bn):
    """x (B, C, M, ns) -> (B, C_out, M, ns)."""
    step = 3 if bn else 2
    for l in range(n_layers):
        w = W[f"{prefix}{l * step}.weight"]
        w = w.reshape(w.shape[0], w.shape[1])
        x = np.einsum("oc,bcmn->bomn", w, x, optimize=True).astype(F32)
        if bn:
            p = f"{prefix}{l * step + 1}."
            inv = (F32(1) / np.sqrt(W[p + "running_var"] + F32(BN_EPS))).astype(F32)
            x = ((x - W[p + "running_mean"][None, :, None, None]) * inv[None, :, None, None] * W[p + "weight"][None, :, None, None]
                 + W[p + "bias"][None, :, None, None]).astype(F32)
        else:
            x = (x + W[f"{prefix}{l * step}.bias"][None, :, None, None]).astype(F32)
        x = np.maximum(x, 0)
    return x


def query_and_group(xyz, new_xyz, features, radius, nsample, use_xyz):
    idx = o.ball_query(radius, nsample, xyz, new_xyz)
    g_xyz = o.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx)
    g_xyz = (g_xyz - new_xyz.transpose(0, 2, 1)[:, :, :, None]).astype(F32)
    if features is None:
        assert use_xyz
        return g_xyz
    g_f = o.group_points(features, idx)
    return np.concatenate([g_xyz, g_f], axis=1) if use_xyz else g_f


def group_all(xyz, features, use_xyz):
    g_xyz = xyz.transpose(0, 2, 1)[:, :, None, :]
    if features is None:
        return np.ascontiguousarray(g_xyz)
    g_f = features[:, :, None, :]
    return np.concatenate([g_xyz, g_f], axis=1) if use_xyz else np.ascontiguousarray(g_f)


def sa_module_forward(W, xyz, features, n_layers, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, prefix="mlps.0."):
    """-> (new_xyz (B,npoint,3) or None, new_features (B, C_out, npoint or 1))."""
    new_xyz = None
    if npoint is not None:
        picked = o.furthest_point_sampling(xyz, npoint)
        new_xyz = np.ascontiguousarray(o.gather_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), picked).transpose(0, 2, 1))
        g = query_and_group(xyz, new_xyz, features, radius, nsample, use_xyz)
    else:
        g = group_all(xyz, features, use_xyz)
    y = shared_mlp(g.astype(F32), W, prefix, n_layers, bn)
    return new_xyz, y.max(axis=3)


def fp_module_forward(W, unknown, known, unknow_feats, known_feats, n_layers, bn=True, prefix="mlp."):
    d, idx = o.three_nn(unknown, known)             # ThreeNN.forward returns sqrt(dist2) (pointnet2_utils.py:124-125)
    rec = (F32(1.0) / (d + F32(1e-8))).astype(F32)
    w = (rec / rec.sum(axis=2, keepdims=True, dtype=F32)).astype(F32)
    f = o.three_interpolate(known_feats, idx, w)
    if unknow_feats is not None:
        f = np.concatenate([f, unknow_feats], axis=1)
    return shared_mlp(f[:, :, :, None].astype(F32), W, prefix, n_layers, bn)[:, :, :, 0]

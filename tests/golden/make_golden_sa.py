#!/usr/bin/env python
"""Golden vectors for the PointNet++ set-abstraction / feature-propagation MODULES.

    python tests/golden/make_golden_sa.py          # dev container only; writes tests/golden/sa_*.npz, samsg_*.npz, fp_*.npz, satrain_*.npz, fptrain_*.npz

Runs the REFERENCE's own Python classes (pointnet2_ops_lib/pointnet2_ops/pointnet2_{utils,modules}.py:
QueryAndGroup, GroupAll, PointnetSAModule, PointnetFPModule, build_shared_mlp) on CPU tensors in eval mode.  Their CUDA
extension ``pointnet2_ops._ext`` cannot be built here (nvcc absent), so it is replaced by a shim over the C oracle
(oracle/pointnet2.c) — the fixtures therefore pin the Python-level composition (centre subtraction, channel order of
[xyz | features], Conv2d + BatchNorm2d(eval) + ReLU stack, max over the neighbourhood, inverse-distance weights),
which is exactly what the fused HIP kernel restates; the per-op semantics are covered by test_oracle_pointnet2_cpu.py.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("DFX_REFERENCE_ROOT", "/root/reference")

from oracle import pointnet2 as o  # noqa: E402

F32 = np.float32


def install_ext_shim():
    ext = types.ModuleType("pointnet2_ops._ext")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    n = lambda x: x.detach().cpu().numpy()
    ext.furthest_point_sampling = lambda xyz, npoint: t(o.furthest_point_sampling(n(xyz), npoint))
    ext.gather_points = lambda feats, idx: t(o.gather_points(n(feats), n(idx)))
    ext.gather_points_grad = lambda g, idx, N: t(o.gather_points_grad(n(g), n(idx), N))
    ext.ball_query = lambda new_xyz, xyz, radius, nsample: t(o.ball_query(radius, nsample, n(xyz), n(new_xyz)))
    ext.group_points = lambda feats, idx: t(o.group_points(n(feats), n(idx)))
    ext.group_points_grad = lambda g, idx, N: t(o.group_points_grad(n(g), n(idx), N))

    def three_nn(unknown, known):
        d2, idx = o.three_nn_dist2(n(unknown), n(known))   # _ext returns squared distances
        return t(d2), t(idx)

    ext.three_nn = three_nn
    ext.three_interpolate = lambda feats, idx, w: t(o.three_interpolate(n(feats), n(idx), n(w)))
    ext.three_interpolate_grad = lambda g, idx, w, m: t(o.three_interpolate_grad(n(g), n(idx), n(w), m))
    sys.modules["pointnet2_ops._ext"] = ext
    sys.path.insert(0, os.path.join(REF, "pointnet2_ops_lib"))
    for k in [k for k in sys.modules if k == "pointnet2_ops" or k.startswith("pointnet2_ops.p")]:
        del sys.modules[k]
    import pointnet2_ops.pointnet2_modules as pm  # the reference's own file
    assert pm.__file__.startswith(REF), pm.__file__
    return pm


def randomize(module, rng):
    """Random conv weights / BN affine + running statistics so that every term of the folded form is exercised."""
    sd = module.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, size=tuple(v.shape))
        elif k.endswith("running_mean") or k.endswith("bias"):
            a = rng.uniform(-0.2, 0.2, size=tuple(v.shape))
        elif v.dim() == 1:
            a = rng.uniform(0.8, 1.2, size=tuple(v.shape))
        else:
            bound = 1.0 / np.sqrt(v.shape[1])
            a = rng.uniform(-bound, bound, size=tuple(v.shape))
        sd[k] = torch.from_numpy(a.astype(F32))
    module.load_state_dict(sd)
    return {k: v.numpy().copy() for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}


def gen_sa(pm, tag, B, N, C, mlp, npoint, radius, nsample, bn, use_xyz, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    feats = rng.standard_normal((B, C, N)).astype(F32) if C else None
    mod = pm.PointnetSAModule(mlp=list(mlp), npoint=npoint, radius=radius, nsample=nsample, bn=bn, use_xyz=use_xyz).eval()
    W = randomize(mod, rng)
    with torch.no_grad():
        new_xyz, new_feats = mod(torch.from_numpy(xyz), None if feats is None else torch.from_numpy(feats))
    out = dict(xyz=xyz, new_features=new_feats.numpy().astype(F32), mlp=np.array(mlp), npoint=np.array(-1 if npoint is None else npoint),
               radius=np.array(0.0 if radius is None else radius, F32), nsample=np.array(-1 if nsample is None else nsample),
               bn=np.array(int(bn)), use_xyz=np.array(int(use_xyz)))
    if feats is not None:
        out["features"] = feats
    if new_xyz is not None:
        out["new_xyz"] = new_xyz.numpy().astype(F32)
    out.update({"w." + k: v for k, v in W.items()})
    np.savez_compressed(os.path.join(HERE, f"sa_{tag}.npz"), **out)
    print("wrote sa_" + tag, new_feats.shape, float(new_feats.abs().max()))


def gen_sa_msg(pm, tag, B, N, C, mlps, npoint, radii, nsamples, bn, use_xyz, seed):
    """PointnetSAModuleMSG (pointnet2_modules.py:77-115): one FPS, one grouper + shared MLP per radius, channel concat (:74)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    feats = rng.standard_normal((B, C, N)).astype(F32) if C else None
    mod = pm.PointnetSAModuleMSG(npoint=npoint, radii=list(radii), nsamples=list(nsamples), mlps=[list(m) for m in mlps], bn=bn, use_xyz=use_xyz).eval()
    W = randomize(mod, rng)
    with torch.no_grad():
        new_xyz, new_feats = mod(torch.from_numpy(xyz), None if feats is None else torch.from_numpy(feats))
    out = dict(xyz=xyz, new_xyz=new_xyz.numpy().astype(F32), new_features=new_feats.numpy().astype(F32), npoint=np.array(npoint),
               radii=np.array(radii, F32), nsamples=np.array(nsamples), bn=np.array(int(bn)), use_xyz=np.array(int(use_xyz)), n_scales=np.array(len(mlps)))
    for i, m in enumerate(mlps):
        out[f"mlp{i}"] = np.array(m)
    if feats is not None:
        out["features"] = feats
    out.update({"w." + k: v for k, v in W.items()})
    np.savez_compressed(os.path.join(HERE, f"samsg_{tag}.npz"), **out)
    print("wrote samsg_" + tag, new_feats.shape, float(new_feats.abs().max()))


def _train_step(mod, inputs, rng):
    """train() forward + backward of a reference module on CPU: out, a random upstream gradient, parameter gradients, input gradients,
    BatchNorm running statistics behind the forward."""
    mod.train()
    W0 = {k: v.numpy().copy() for k, v in mod.state_dict().items() if not k.endswith("num_batches_tracked")}
    res = mod(*inputs)
    out = res[1] if isinstance(res, tuple) else res
    gout = rng.standard_normal(tuple(out.shape)).astype(F32)
    out.backward(torch.from_numpy(gout))
    d = {"new_features": out.detach().numpy().astype(F32), "gout": gout}
    d.update({"w." + k: v for k, v in W0.items()})
    d.update({"g." + k: p.grad.numpy().astype(F32) for k, p in mod.named_parameters()})
    d.update({"after." + k: v.numpy().copy() for k, v in mod.state_dict().items() if "running_" in k})
    return res, d


def gen_sa_train(pm, tag, B, N, C, mlps, npoint, radii, nsamples, bn, use_xyz, seed):
    """PointnetSAModule(MSG) in train() mode (BatchNorm2d on batch statistics, autograd through grouping / conv / max_pool2d) — the reference's Python
    classes over the C-oracle shim, whose *_grad entry points serve the backward."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    feats = rng.standard_normal((B, C, N)).astype(F32) if C else None
    mod = pm.PointnetSAModuleMSG(npoint=npoint, radii=list(radii), nsamples=list(nsamples), mlps=[list(m) for m in mlps], bn=bn, use_xyz=use_xyz)
    randomize(mod, rng)
    tx = torch.from_numpy(xyz).requires_grad_(True)
    tf = None if feats is None else torch.from_numpy(feats).requires_grad_(True)
    res, out = _train_step(mod, (tx, tf), rng)
    out.update(dict(xyz=xyz, npoint=np.array(-1 if npoint is None else npoint), radii=np.array([0.0 if r is None else r for r in radii], F32),
                    nsamples=np.array([-1 if n is None else n for n in nsamples]), bn=np.array(int(bn)), use_xyz=np.array(int(use_xyz)), n_scales=np.array(len(mlps))))
    if res[0] is not None:
        out["new_xyz"] = res[0].detach().numpy().astype(F32)
    if tx.grad is not None:
        out["d_xyz"] = tx.grad.numpy().astype(F32)
    for i, m in enumerate(mlps):
        out[f"mlp{i}"] = np.array(m)
    if feats is not None:
        out["features"], out["d_features"] = feats, tf.grad.numpy().astype(F32)
    np.savez_compressed(os.path.join(HERE, f"satrain_{tag}.npz"), **out)
    print("wrote satrain_" + tag, out["new_features"].shape, float(np.abs(out["new_features"]).max()))


def gen_fp_train(pm, tag, B, n, m, C1, C2, mlp, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    unknown = rng.uniform(-1, 1, size=(B, n, 3)).astype(F32)
    known = rng.uniform(-1, 1, size=(B, m, 3)).astype(F32)
    uf = rng.standard_normal((B, C1, n)).astype(F32)
    kf = rng.standard_normal((B, C2, m)).astype(F32)
    mod = pm.PointnetFPModule(mlp=list(mlp))
    randomize(mod, rng)
    tu, tk = torch.from_numpy(uf).requires_grad_(True), torch.from_numpy(kf).requires_grad_(True)
    _, out = _train_step(mod, (torch.from_numpy(unknown), torch.from_numpy(known), tu, tk), rng)
    out.update(dict(unknown=unknown, known=known, unknow_feats=uf, known_feats=kf, mlp=np.array(mlp), d_unknow_feats=tu.grad.numpy().astype(F32),
                    d_known_feats=tk.grad.numpy().astype(F32)))
    np.savez_compressed(os.path.join(HERE, f"fptrain_{tag}.npz"), **out)
    print("wrote fptrain_" + tag, out["new_features"].shape)


def gen_fp(pm, tag, B, n, m, C1, C2, mlp, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    unknown = rng.uniform(-1, 1, size=(B, n, 3)).astype(F32)
    known = rng.uniform(-1, 1, size=(B, m, 3)).astype(F32)
    uf = rng.standard_normal((B, C1, n)).astype(F32)
    kf = rng.standard_normal((B, C2, m)).astype(F32)
    mod = pm.PointnetFPModule(mlp=list(mlp)).eval()
    W = randomize(mod, rng)
    with torch.no_grad():
        out_f = mod(*map(torch.from_numpy, (unknown, known, uf, kf)))
    out = dict(unknown=unknown, known=known, unknow_feats=uf, known_feats=kf, new_features=out_f.numpy().astype(F32), mlp=np.array(mlp))
    out.update({"w." + k: v for k, v in W.items()})
    np.savez_compressed(os.path.join(HERE, f"fp_{tag}.npz"), **out)
    print("wrote fp_" + tag, out_f.shape)


def main():
    torch.manual_seed(0)
    pm = install_ext_shim()
    # PointNet2SSG's three layers at reduced size (python/difffacto/models/encoders/pointnet2.py:18-45)
    gen_sa(pm, "ssg1_small", B=2, N=256, C=4, mlp=[4, 64, 64, 128], npoint=64, radius=0.4, nsample=64, bn=True, use_xyz=True, seed=41)
    gen_sa(pm, "ssg2_small", B=2, N=96, C=128, mlp=[128, 128, 128, 256], npoint=24, radius=0.7, nsample=64, bn=True, use_xyz=True, seed=42)
    gen_sa(pm, "groupall", B=3, N=40, C=16, mlp=[16, 32, 64], npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, seed=43)
    gen_sa(pm, "nobn_noxyz_ragged", B=2, N=100, C=5, mlp=[5, 24, 40], npoint=10, radius=0.5, nsample=20, bn=False, use_xyz=False, seed=44)
    gen_sa(pm, "xyz_only", B=1, N=128, C=0, mlp=[0, 16, 32], npoint=16, radius=0.6, nsample=16, bn=True, use_xyz=True, seed=45)
    gen_fp(pm, "small", B=2, n=50, m=20, C1=6, C2=10, mlp=[16, 32, 24], seed=46)
    # PointNet2MSG's first two layers at reduced size (python/difffacto/models/encoders/pointnet2.py:88-112): three radii, nsample 16 / 32 / 128,
    # channel concat 64 + 128 + 128 -> second layer's input
    gen_sa_msg(pm, "msg1_small", B=2, N=320, C=3, mlps=[[3, 32, 32, 64], [3, 64, 64, 128], [3, 64, 96, 128]], npoint=48, radii=[0.2, 0.4, 0.8],
               nsamples=[16, 32, 128], bn=True, use_xyz=True, seed=47)
    gen_sa_msg(pm, "msg2_small", B=2, N=64, C=320, mlps=[[320, 64, 64, 128], [320, 128, 128, 256]], npoint=16, radii=[0.5, 1.0],
               nsamples=[32, 64], bn=True, use_xyz=True, seed=48)
    # train() mode (round 6): batch-statistics BatchNorm + autograd, the layers' native training kernels (dfx_shared_mlp_train_*) are checked against these
    gen_sa_train(pm, "ssg_small", B=3, N=160, C=4, mlps=[[4, 32, 32, 64]], npoint=24, radii=[0.5], nsamples=[16], bn=True, use_xyz=True, seed=51)
    gen_sa_train(pm, "msg_small", B=2, N=128, C=6, mlps=[[6, 16, 32], [6, 32, 48]], npoint=16, radii=[0.4, 0.8], nsamples=[8, 24], bn=True, use_xyz=True, seed=52)
    gen_sa_train(pm, "nobn_groupall", B=2, N=40, C=5, mlps=[[5, 24, 40]], npoint=None, radii=[None], nsamples=[None], bn=False, use_xyz=False, seed=53)
    gen_fp_train(pm, "small", B=2, n=50, m=20, C1=6, C2=10, mlp=[16, 32, 24], seed=54)


if __name__ == "__main__":
    main()

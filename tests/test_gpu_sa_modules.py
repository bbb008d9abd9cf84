"""GPU parity of the fused / general set-abstraction and feature-propagation paths through
the C-ABI (dfx_shared_mlp_create, dfx_sa_forward_f32, dfx_fp_forward_f32), driven through the module mirrors:

* vs goldens produced by the reference's own Python classes (tests/golden/sa_*.npz, fp_*.npz);
* fused vs general path at the PointNet2SSG shapes (SA1, SA2) and the general path for SA3 (1024 outputs);
* vs the module's own torch layers (train-mode code path evaluated in eval mode, i.e. conv + BN(running) + ReLU + max).

Tolerance: fp32 MFMA vs fp32 conv differ by summation order and by folding BN into the weights: 2e-5 x max(1, |ref|).
FPS / ball-query indices are bit-exact (asserted through new_xyz).
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5


def _close(a, ref, tol=TOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else ref
    assert a.shape == ref.shape, (a.shape, ref.shape)
    err = float(np.abs(a - ref).max())
    assert err <= tol * max(1.0, float(np.abs(ref).max())), err


def _load(module, g):
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    for k in module.state_dict():
        if k.endswith("num_batches_tracked"):
            sd[k] = module.state_dict()[k]
    module.load_state_dict(sd)
    return module.cuda().eval()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "sa_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("force_general", [False, True])
def test_sa_module_matches_reference_golden(path, force_general):
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule
    g = np.load(path)
    npoint = int(g["npoint"])
    mod = PointnetSAModule(mlp=[int(v) for v in g["mlp"]], npoint=None if npoint < 0 else npoint,
                           radius=None if npoint < 0 else float(g["radius"]), nsample=None if npoint < 0 else int(g["nsample"]),
                           bn=bool(g["bn"]), use_xyz=bool(g["use_xyz"]))
    mod = _load(mod, g)
    xyz = torch.from_numpy(g["xyz"]).cuda()
    feats = torch.from_numpy(g["features"]).cuda() if "features" in g.files else None
    with torch.no_grad():
        new_xyz = mod._centres(xyz)
        out = mod._forward_native(0, xyz, new_xyz, feats, force_general=force_general)
        nx2, out2 = mod(xyz, feats)
    if npoint >= 0:
        assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    _close(out, g["new_features"])
    if not force_general:
        _close(out2, g["new_features"])


@pytest.mark.parametrize("tag", ["msg1_small", "msg2_small"])
def test_sa_msg_module_matches_reference_golden(tag):
    """Multi-scale grouping (PointnetSAModuleMSG, pointnet2_modules.py:77-115; PointNet2MSG's layers, models/encoders/pointnet2.py:88-112):
    ONE furthest-point sampling, then per radius ball query -> group -> shared MLP -> max (dfx_sa_forward_f32 per scale) and the channel
    concat — against the reference's own class (tests/golden/samsg_*.npz, make_golden_sa.py).  Also each scale's general path."""
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleMSG
    g = np.load(os.path.join(GOLDEN, f"samsg_{tag}.npz"))
    S = int(g["n_scales"])
    mlps = [[int(v) for v in g[f"mlp{i}"]] for i in range(S)]
    mod = PointnetSAModuleMSG(npoint=int(g["npoint"]), radii=[float(r) for r in g["radii"]], nsamples=[int(n) for n in g["nsamples"]],
                              mlps=[list(m) for m in mlps], bn=bool(g["bn"]), use_xyz=bool(g["use_xyz"]))
    mod = _load(mod, g)
    xyz = torch.from_numpy(g["xyz"]).cuda()
    feats = torch.from_numpy(g["features"]).cuda() if "features" in g.files else None
    with torch.no_grad():
        new_xyz, out = mod(xyz, feats)
        per_scale = [mod._forward_native(k, xyz, new_xyz, feats, force_general=True) for k in range(S)]
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])            # FPS + gather: bit-exact
    assert out.shape[1] == sum(m[-1] for m in mlps)
    _close(out, g["new_features"])
    _close(torch.cat(per_scale, dim=1), g["new_features"])
    print(f"samsg_{tag}: {S} scales -> {tuple(out.shape)}, max err {float(np.abs(out.cpu().numpy() - g['new_features']).max()):.2e} (|ref| {float(np.abs(g['new_features']).max()):.2f})")


def test_fp_module_matches_reference_golden():
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetFPModule
    g = np.load(os.path.join(GOLDEN, "fp_small.npz"))
    mod = _load(PointnetFPModule(mlp=[int(v) for v in g["mlp"]]), g)
    c = lambda k: torch.from_numpy(g[k]).cuda()
    with torch.no_grad():
        out = mod(c("unknown"), c("known"), c("unknow_feats"), c("known_feats"))
    _close(out, g["new_features"])
    with torch.no_grad():   # known is None: (B, C2, 1) features broadcast
        kf1 = c("known_feats")[:, :, :1].contiguous()
        out_n = mod(c("unknown"), None, c("unknow_feats"), kf1)
    with torch.no_grad():   # the torch layers on the broadcast features
        stacked = torch.cat([kf1.expand(-1, -1, c("unknown").shape[1]), c("unknow_feats")], dim=1)
        ref_n = mod.mlp(stacked.unsqueeze(-1)).squeeze(-1)
    _close(out_n, ref_n.detach())


def _torch_layers(mod, xyz, feats):
    """The module's own torch layers (nn.Conv2d / nn.BatchNorm2d / ReLU + amax) on the grouper's output: the reference's forward, spelled out — since
    round 6 `mod(...)` itself reaches them only for stacks the native training kernels do not serve."""
    new_xyz = mod._centres(xyz)
    return new_xyz, torch.cat([mlp(grouper(xyz, new_xyz, feats)).amax(dim=3) for grouper, mlp in zip(mod.groupers, mod.mlps)], dim=1)


def _randomize(mod, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = mod.state_dict()
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
            a = rng.uniform(-1, 1, size=tuple(v.shape)) / np.sqrt(v.shape[1])
        sd[k] = torch.from_numpy(a.astype(np.float32))
    mod.load_state_dict(sd)
    return mod.cuda().eval()


def test_pointnet2_ssg_shapes_fused_vs_general_vs_torch():
    """SA1 / SA2 / SA3 of PointNet2SSG (python/difffacto/models/encoders/pointnet2.py:18-45) at B = 8, N = 2048."""
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule
    from difffacto_amd import _ffi
    torch.manual_seed(0)
    B = 8
    rng = np.random.Generator(np.random.PCG64(9))
    xyz = torch.from_numpy(rng.uniform(-1, 1, size=(B, 2048, 3)).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.standard_normal((B, 4, 2048)).astype(np.float32)).cuda()
    sa = [_randomize(PointnetSAModule(npoint=512, radius=0.2, nsample=64, mlp=[4, 64, 64, 128]), 1),
          _randomize(PointnetSAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256]), 2),
          _randomize(PointnetSAModule(mlp=[256, 256, 512, 1024]), 3)]
    fused_expected = [1, 1, 0]
    for mod, fe in zip(sa, fused_expected):
        with torch.no_grad():
            new_xyz = mod._centres(xyz)
            out_f = mod._forward_native(0, xyz, new_xyz, feats)
            assert _ffi.lib().dfx_shared_mlp_is_fused(mod._native(0).handle()) == fe
            out_g = mod._forward_native(0, xyz, new_xyz, feats, force_general=True)
        with torch.no_grad():   # the torch layers, eval-mode BN
            nx_t, out_t = _torch_layers(mod, xyz, feats)
        _close(out_f, out_g)
        _close(out_f, out_t.detach())
        if new_xyz is not None:
            assert torch.equal(new_xyz, nx_t)
        xyz, feats = new_xyz, out_f
    assert tuple(feats.shape) == (B, 1024, 1)


def test_native_mlp_tracks_parameter_updates():
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule
    mod = _randomize(PointnetSAModule(npoint=16, radius=0.5, nsample=8, mlp=[0, 8, 16]), 5)
    xyz = torch.rand(2, 64, 3, device="cuda") * 2 - 1
    with torch.no_grad():
        _, a = mod(xyz, None)
        mod.mlps[0][0].weight.mul_(2.0)
        _, b = mod(xyz, None)
    with torch.no_grad():
        _, ref = _torch_layers(mod, xyz, None)
    with torch.enable_grad():   # eval() under autograd: the training kernels on the running statistics — the same numbers
        _, ref2 = mod(xyz, None)
    _close(b, ref.detach())
    _close(ref2.detach(), ref.detach())
    assert not torch.allclose(a, b)


@pytest.mark.parametrize("B,N,npoint,nsample,mlp", [
    (1, 300, 3, 64, [4, 64, 64, 128]),       # 6 tiles: the last workgroup has two spare wavefronts (they take part in its barriers)
    (2, 400, 37, 96, [4, 64, 64, 128]),      # 3 tiles per centre: a centre straddles workgroups -> atomic max into the zero-filled output
    (2, 400, 50, 40, [4, 64, 64, 128]),      # 2 tiles per centre, the second one mostly padding
    (2, 400, 33, 128, [4, 128, 128, 256]),   # 4 tiles per centre = one workgroup
    (3, 256, 21, 16, [4, 32, 32, 64]),       # run-time tile counts (1, 1), one tile per centre
    (2, 256, 19, 64, [4, 64, 96, 64]),       # run-time tile counts (2, 3): odd number of input tiles -> the fragment sets swap by a copy
    (2, 256, 30, 64, [4, 64, 128]),          # two layers
    (2, 256, 30, 32, [4, 128, 256]),         # two layers, 4 input tiles
    (1, 200, 10, 70, [9, 64, 64, 100]),      # 12 + 3 input channels (two K units), ragged output width
])
def test_fused_set_abstraction_variants_vs_general(B, N, npoint, nsample, mlp):
    """Every code path of the fused kernel's output stage and weight pipeline (compile-time / run-time tile counts, LDS merge of a
    centre's tiles / atomics, spare wavefronts) against the layer-by-layer path of the same library and the torch layers."""
    from difffacto_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule
    from difffacto_amd import _ffi
    rng = np.random.Generator(np.random.PCG64(B * 1000 + nsample))
    xyz = torch.from_numpy(rng.uniform(-1, 1, size=(B, N, 3)).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.standard_normal((B, mlp[0], N)).astype(np.float32)).cuda()
    mod = _randomize(PointnetSAModule(npoint=npoint, radius=0.6, nsample=nsample, mlp=list(mlp)), nsample)
    with torch.no_grad():
        new_xyz = mod._centres(xyz)
        out_f = mod._forward_native(0, xyz, new_xyz, feats)
        assert _ffi.lib().dfx_shared_mlp_is_fused(mod._native(0).handle()) == 1
        out_g = mod._forward_native(0, xyz, new_xyz, feats, force_general=True)
        again = mod._forward_native(0, xyz, new_xyz, feats)
    with torch.no_grad():
        _, out_t = _torch_layers(mod, xyz, feats)
    assert torch.equal(out_f, again)
    _close(out_f, out_g)
    _close(out_f, out_t.detach())


# ---- train() mode (round 6): the shared MLP with batch-statistics BatchNorm, the max and their gradients on libdfx (dfx_shared_mlp_train_*) against goldens the
# reference's own classes produced under autograd (tests/golden/satrain_*.npz, fptrain_*.npz: make_golden_sa.py).  fp32 sums in another order than torch's,
# a BatchNorm in between: 2e-4 x max(1, |ref|) on outputs / running statistics, gradients 5e-4 of their max-abs (measured ~1e-5). ----
def _load_train(module, g):
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    for k in module.state_dict():
        if k.endswith("num_batches_tracked"):
            sd[k] = module.state_dict()[k]
    module.load_state_dict(sd)
    return module.cuda().train()


def _check_train(mod, g, out, extra):
    from difffacto_amd import _ffi
    _close(out, g["new_features"], 2e-4)
    out.backward(torch.from_numpy(g["gout"]).cuda())
    worst, at = 0.0, None
    grads = {k: p.grad for k, p in mod.named_parameters()}
    grads.update(extra())
    n = 0
    for k in g.files:
        if k.startswith("g.") or k.startswith("d_"):
            ref = g[k]
            got = grads[k[2:] if k.startswith("g.") else k].detach().cpu().numpy()
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            e = float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
            assert e <= 5e-4, (k, e)
            if e > worst:
                worst, at = e, k
            n += 1
    sd = mod.state_dict()
    for k in g.files:
        if k.startswith("after."):
            _close(sd[k[6:]], g[k], 2e-4)
    for name, buf in mod.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(buf) == 1, name
    return n, worst, at


@pytest.mark.parametrize("tag", ["ssg_small", "msg_small", "nobn_groupall"])
def test_sa_module_train_mode_is_native_and_matches_reference_autograd(tag, monkeypatch):
    from difffacto_amd.pointnet2_ops import pointnet2_modules as pm
    g = np.load(os.path.join(GOLDEN, f"satrain_{tag}.npz"))
    S = int(g["n_scales"])
    npoint = int(g["npoint"])
    mod = pm.PointnetSAModuleMSG(npoint=None if npoint < 0 else npoint, radii=[None if npoint < 0 else float(r) for r in g["radii"]],
                                 nsamples=[None if npoint < 0 else int(n) for n in g["nsamples"]], mlps=[[int(v) for v in g[f"mlp{i}"]] for i in range(S)],
                                 bn=bool(g["bn"]), use_xyz=bool(g["use_xyz"]))
    mod = _load_train(mod, g)
    for seq in mod.mlps:   # the module's torch layers must not run: the native path or nothing
        for layer in seq:
            if not isinstance(layer, torch.nn.ReLU):
                monkeypatch.setattr(layer, "forward", lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch layer used in train mode")))
    xyz = torch.from_numpy(g["xyz"]).cuda().requires_grad_(True)
    feats = torch.from_numpy(g["features"]).cuda().requires_grad_(True) if "features" in g.files else None
    new_xyz, out = mod(xyz, feats)
    if npoint >= 0:
        assert np.array_equal(new_xyz.detach().cpu().numpy(), g["new_xyz"])
    n, worst, at = _check_train(mod, g, out, lambda: {"d_xyz": xyz.grad, **({} if feats is None else {"d_features": feats.grad})})
    print(f"satrain_{tag}: out {tuple(out.shape)}, {n} gradient tensors vs the reference's autograd, worst {worst:.1e} of max-abs ({at})")


def test_fp_module_train_mode_is_native_and_matches_reference_autograd(monkeypatch):
    from difffacto_amd.pointnet2_ops import pointnet2_modules as pm
    g = np.load(os.path.join(GOLDEN, "fptrain_small.npz"))
    mod = _load_train(pm.PointnetFPModule(mlp=[int(v) for v in g["mlp"]]), g)
    for layer in mod.mlp:
        if not isinstance(layer, torch.nn.ReLU):
            monkeypatch.setattr(layer, "forward", lambda *a, **k: (_ for _ in ()).throw(AssertionError("torch layer used in train mode")))
    c = lambda k: torch.from_numpy(g[k]).cuda()
    uf, kf = c("unknow_feats").requires_grad_(True), c("known_feats").requires_grad_(True)
    out = mod(c("unknown"), c("known"), uf, kf)
    n, worst, at = _check_train(mod, g, out, lambda: {"d_unknow_feats": uf.grad, "d_known_feats": kf.grad})
    print(f"fptrain_small: out {tuple(out.shape)}, {n} gradient tensors, worst {worst:.1e} of max-abs ({at})")


def test_shared_mlp_train_against_torch_layers_at_pointnet2_ssg_sizes():
    """SA1 of PointNet2SSG (mlp [3 + 3, 64, 64, 128], 512 centres x 32 neighbours, B = 4) and a 1024-wide layer: the native training op against the
    module's own torch layers (cuDNN-free: Conv2d / BatchNorm2d on the same device) — output, running statistics, every gradient."""
    from difffacto_amd.pointnet2_ops import pointnet2_modules as pm
    for spec, B, M, ns, pool, train in (([6, 64, 64, 128], 4, 512, 32, True, True), ([259, 256, 512, 1024], 2, 1, 128, True, True), ([131, 128, 128], 2, 300, 1, False, True),
                                        ([6, 64, 64, 128], 2, 256, 16, True, False)):   # (the last: eval() under autograd — BatchNorm on its running statistics)
        ref = _randomize(pm.build_shared_mlp(list(spec), bn=True), 7).train(train)
        mine = _randomize(pm.build_shared_mlp(list(spec), bn=True), 7).train(train)
        x = torch.randn(B, spec[0], M, ns, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        yr = ref(xr).amax(dim=3) if pool else ref(xr)
        ym = pm.shared_mlp_train(mine, xm, pool=pool)
        gout = torch.randn_like(yr)
        yr.backward(gout)
        ym.backward(gout)
        _close(ym, yr.detach(), 2e-4)
        worst = float((xm.grad - xr.grad).abs().max() / xr.grad.abs().max())
        for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
            worst = max(worst, float((p.grad - q.grad).abs().max() / q.grad.abs().max().clamp_min(1e-30)))
        for (k, b), (_, c) in zip(mine.named_buffers(), ref.named_buffers()):
            if "running" in k:
                _close(b, c, 2e-4)
        print(f"shared MLP {'train' if train else 'eval + grad'} {spec} x (B={B}, M={M}, ns={ns}): worst gradient error {worst:.1e} of max-abs")
        assert worst <= 1e-3, worst

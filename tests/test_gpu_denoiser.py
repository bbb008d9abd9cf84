"""GPU parity of the hot path: libdfx denoiser / p_sample / persistent chain (through the C-ABI) vs
(a) the golden vectors produced by the reference's own Python model and (b) the numpy oracle on
seeded inputs.  Tolerances are stated per precision."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from difffacto_amd import synth  # noqa: E402
from oracle import denoiser as odn  # noqa: E402
from oracle import diffusion as odf  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

# BASELINE.md config 1 gate for the HIP fp32 path is 1e-3; measured error is ~1e-5.
TOL_F32_EPS = 1e-4
TOL_F32_CHAIN = 1e-3
# bf16 operands / fp32 accumulate, packed-fp16 polynomial GELU: |eps| <= 1.9, measured max-abs 2.3e-3 for one evaluation
# (N = 2048 golden; rms 6.6e-4) -> gate at <= 3x measured
TOL_BF16_EPS = 6e-3
# T = 100 chain, bf16 vs exact fp32 on identical noise: measured max-abs 1.17e-3 (part sigma 0.22) -> gate at 3x
TOL_BF16_CHAIN_T100 = 3.5e-3
# the pipelined and the direct bf16 kernels share the bf16 operands and differ in the GELU / LayerNorm formulation
TOL_PIPE_VS_DIRECT = 4e-3   # measured 1.3e-3 (one evaluation), 9e-5 (T = 3 chain)


@pytest.fixture(scope="module")
def W():
    return synth.make_denoiser_weights(seed=0)


def _engine(W, T, prec):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.engine import DenoiserEngine
    return DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision=prec)


@pytest.fixture(scope="module")
def eng10(W):
    return _engine(W, 10, "f32")


@pytest.fixture(scope="module")
def eng10_bf16(W):
    return _engine(W, 10, "bf16")


def _prep(eng, g):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return eng.prepare_shapes(t(g["part_code"]), t(g["mean"]), t(np.exp(g["logvar"]).astype(np.float32)), t(g["valid"]))


@pytest.mark.parametrize("T", [10, 100, 1000])
def test_tables_bit_exact_vs_reference(W, T):
    eng = _engine(W, T, "f32")
    g = np.load(os.path.join(GOLDEN, f"tables_T{T}.npz"))
    tabs = eng.tables()
    for name in g.files:
        assert np.array_equal(tabs[name], g[name]), name


@pytest.mark.parametrize("tag", ["B2_N128_mixed", "B2_N128_allvalid", "B1_N2048"])
def test_eps_f32_vs_reference_golden(eng10, tag):
    g = np.load(os.path.join(GOLDEN, f"denoiser_eps_{tag}.npz"))
    ctx = _prep(eng10, g)
    for t in g["ts"]:
        eps = eng10.eps(ctx, torch.from_numpy(g["x"]), torch.from_numpy(g["seg"]), int(t)).cpu().numpy()
        err = np.abs(eps - g[f"eps_t{int(t)}"]).max()
        assert err < TOL_F32_EPS, (tag, t, err)


@pytest.mark.parametrize("tag", ["B2_N128_mixed", "B1_N2048"])
def test_eps_bf16_vs_reference_golden(eng10_bf16, tag):
    g = np.load(os.path.join(GOLDEN, f"denoiser_eps_{tag}.npz"))
    ctx = _prep(eng10_bf16, g)
    for t in g["ts"]:
        eps = eng10_bf16.eps(ctx, torch.from_numpy(g["x"]), torch.from_numpy(g["seg"]), int(t)).cpu().numpy()
        ref = g[f"eps_t{int(t)}"]
        err = np.abs(eps - ref).max()
        print(f"bf16 eps {tag} t={t}: max-abs {err:.3e} rms {np.sqrt(((eps - ref) ** 2).mean()):.3e} (|ref| max {np.abs(ref).max():.2f})")
        assert err < TOL_BF16_EPS, (tag, t, err)


@pytest.mark.parametrize("tag", ["B2_N128_mixed", "B3_N64_allvalid"])
def test_chain_f32_vs_reference_golden(eng10, tag):
    g = np.load(os.path.join(GOLDEN, f"chain_T10_{tag}.npz"))
    ctx = _prep(eng10, g)
    seg = torch.from_numpy(g["seg"])
    ri = int(g["ret_interval"])
    pred, traj = eng10.sample_chain(ctx, seg, x_T_noise=torch.from_numpy(g["x_T_noise"]),
                                    step_noise=torch.from_numpy(g["step_noise"]), ret_interval=ri)
    pred = pred.cpu().numpy()
    assert np.abs(pred - g["decode_pred"]).max() < TOL_F32_CHAIN
    assert np.abs(pred - g["traj"][-1].transpose(0, 2, 1)).max() < TOL_F32_CHAIN
    times = eng10.snapshot_times(ri)
    assert times == [10, 5]
    for k, t in enumerate(times):
        assert np.abs(traj[k].cpu().numpy() - g[f"decode_{t}"]).max() < TOL_F32_CHAIN, t
    # the per-step entry point (generator API) walks the same trajectory
    x = torch.from_numpy(g["traj"][0]).cuda()
    for i, t in enumerate(range(9, -1, -1)):
        x = eng10.p_sample(ctx, x, seg, t, noise=torch.from_numpy(g["step_noise"][i]))
        assert np.abs(x.cpu().numpy() - g["traj"][i + 1]).max() < TOL_F32_CHAIN, t


def test_pred_xstart_output(eng10, W):
    g = np.load(os.path.join(GOLDEN, "chain_T10_B2_N128_mixed.npz"))
    ctx = _prep(eng10, g)
    tb = odf.Tables(10)
    var = np.exp(g["logvar"]).astype(np.float32)
    anchors, variance = odf.gather_params(g["seg"], g["mean"], var)
    cl = [g["part_code"], np.concatenate([g["mean"], var], 1)]
    x = g["traj"][3]
    out = odf.p_sample(tb, W, x, 6, anchors, cl, variance, g["seg"], g["valid"], g["step_noise"][3])
    xs_prev, xstart = eng10.p_sample(ctx, torch.from_numpy(x), torch.from_numpy(g["seg"]), 6,
                                     noise=torch.from_numpy(g["step_noise"][3]), want_xstart=True)
    assert np.abs(xs_prev.cpu().numpy() - out["sample"]).max() < TOL_F32_CHAIN
    assert np.abs(xstart.cpu().numpy() - out["pred_xstart"]).max() < 5e-3   # x0-hat is amplified by sqrt(1/abar - 1)


@pytest.mark.parametrize("B,N,all_valid", [(5, 256, False), (2, 2048, True), (1, 8192, False), (7, 32, False), (3, 100, False), (2, 8, True)])
def test_eps_f32_vs_oracle_seeded(W, B, N, all_valid):
    eng = _engine(W, 100, "f32")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=B * 1000 + N, all_valid=all_valid)
    if B == 7:
        valid[0] = [0, 0, 1, 0]     # single valid key
        valid[1] = [0, 1, 0, 1]
    seg = synth.make_seg_mask(valid, N)
    var = np.exp(logvar).astype(np.float32)
    rng = np.random.default_rng(B + N)
    anchors, variance = odf.gather_params(seg, mean, var)
    x = (np.sqrt(variance) * rng.standard_normal((B, 3, N)).astype(np.float32) + anchors).astype(np.float32)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    for t in (0, 37, 99):
        ref = odn.transformer_net_forward(W, x, np.full((B,), t), [part_code, np.concatenate([mean, var], 1)],
                                          anchors.transpose(0, 2, 1), variance.transpose(0, 2, 1), valid, seg)
        eps = eng.eps(ctx, torch.from_numpy(x), torch.from_numpy(seg), t).cpu().numpy()
        assert np.abs(eps - ref).max() < TOL_F32_EPS, (t, np.abs(eps - ref).max())


def test_point_counts_that_are_not_multiples_of_32(W):
    """The reference takes any N; the kernels take N % 32 == 0 and the host side pads (points are independent): one posterior
    step with pred_xstart, the DDPM chain with snapshots and q_sample at N = 100 against the oracle."""
    T, B, N = 12, 3, 100
    eng = _engine(W, T, "f32")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=41)
    seg = synth.make_seg_mask(valid, N)
    var = np.exp(logvar).astype(np.float32)
    rng = np.random.default_rng(3)
    xT = rng.standard_normal((B, 3, N)).astype(np.float32)
    zs = rng.standard_normal((T, B, 3, N)).astype(np.float32)
    anchors, variance = odf.gather_params(seg, mean, var)
    cctx = [part_code, np.concatenate([mean, var], 1)]
    tb = odf.Tables(T)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    x = (np.sqrt(variance) * xT + anchors).astype(np.float32)
    ref = odf.p_sample(tb, W, x, 7, anchors, cctx, variance, seg, valid, zs[0])
    out, xs = eng.p_sample(ctx, torch.from_numpy(x), torch.from_numpy(seg), 7, noise=torch.from_numpy(zs[0]), want_xstart=True)
    assert out.shape == (B, 3, N) and out.is_contiguous()
    assert np.abs(out.cpu().numpy() - ref["sample"]).max() < TOL_F32_EPS * 10
    assert np.abs(xs.cpu().numpy() - ref["pred_xstart"]).max() < TOL_F32_EPS * 10
    dec = odf.decode(tb, W, anchors, cctx, variance, seg, valid, xT, zs, ret_traj=True, ret_interval=4)
    pred, traj = eng.sample_chain(ctx, torch.from_numpy(seg), x_T_noise=torch.from_numpy(xT), step_noise=torch.from_numpy(zs), ret_interval=4)
    assert pred.shape == (B, N, 3) and traj.shape[1:] == (B, N, 3)
    assert np.abs(pred.cpu().numpy() - dec["pred"]).max() < TOL_F32_CHAIN
    for k, t in enumerate(eng.snapshot_times(4)):
        assert np.abs(traj[k].cpu().numpy() - dec[t]).max() < TOL_F32_CHAIN
    tt = np.array([0, 5, 11])
    q = eng.q_sample(ctx, torch.from_numpy(seg), torch.from_numpy(x), torch.from_numpy(tt), torch.from_numpy(zs[1]))
    assert np.abs(q.cpu().numpy() - odf.q_sample(tb, x, tt, anchors, zs[1], variance)).max() < 1e-5
    pred2, _ = eng.sample_chain(ctx, torch.from_numpy(seg), seed=5)          # in-kernel noise: shape and finiteness
    assert pred2.shape == (B, N, 3) and bool(torch.isfinite(pred2).all())


@pytest.mark.parametrize("N", [96, 288, 480])
def test_pipelined_kernel_partial_tiles_are_bit_identical_to_full_tiles(W, N):
    """The bf16 pipelined kernel works on 256-point tiles of one shape; a partial last tile idles whole wavefronts.  Points are
    independent, so the first N points of a 512-point launch and an N-point launch of the same inputs must agree bit for bit
    (eps, and a T = 6 chain with explicit noise and snapshots)."""
    T, B, NF = 6, 3, 512
    eng = _engine(W, T, "bf16")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=N)
    var = np.exp(logvar).astype(np.float32)
    rng = np.random.default_rng(N)
    seg = rng.integers(0, 4, size=(B, NF)).astype(np.int32)
    seg = np.where(valid[np.arange(B)[:, None], seg] > 0, seg, np.argmax(valid, axis=1)[:, None]).astype(np.int32)   # only present parts
    x = rng.standard_normal((B, 3, NF)).astype(np.float32)
    zs = rng.standard_normal((T, B, 3, NF)).astype(np.float32)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    full = eng.eps(ctx, cu(x), cu(seg), 3)
    part = eng.eps(ctx, cu(x[..., :N]), cu(seg[:, :N]), 3)
    assert torch.equal(part, full[..., :N])
    pf, tf = eng.sample_chain(ctx, cu(seg), x_T_noise=cu(x), step_noise=cu(zs), ret_interval=2)
    pp, tp = eng.sample_chain(ctx, cu(seg[:, :N]), x_T_noise=cu(x[..., :N]), step_noise=cu(zs[..., :N]), ret_interval=2)
    assert torch.equal(pp, pf[:, :N]) and torch.equal(tp, tf[:, :, :N])
    assert bool(torch.isfinite(pp).all())


def test_chain_f32_vs_oracle_T100(W):
    """Shipped schedule length (num_timesteps=100), explicit noise, whole chain in one launch."""
    T, B, N = 100, 2, 64
    eng = _engine(W, T, "f32")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=77)
    seg = synth.make_seg_mask(valid, N)
    var = np.exp(logvar).astype(np.float32)
    rng = np.random.default_rng(9)
    xT = rng.standard_normal((B, 3, N)).astype(np.float32)
    zs = rng.standard_normal((T, B, 3, N)).astype(np.float32)
    anchors, variance = odf.gather_params(seg, mean, var)
    dec = odf.decode(odf.Tables(T), W, anchors, [part_code, np.concatenate([mean, var], 1)], variance, seg, valid, xT, zs,
                     ret_traj=True, ret_interval=10)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    pred, traj = eng.sample_chain(ctx, torch.from_numpy(seg), x_T_noise=torch.from_numpy(xT),
                                  step_noise=torch.from_numpy(zs), ret_interval=10)
    err = np.abs(pred.cpu().numpy() - dec["pred"]).max()
    print("T=100 f32 chain max-abs vs oracle:", err)
    assert err < TOL_F32_CHAIN
    for k, t in enumerate(eng.snapshot_times(10)):
        assert np.abs(traj[k].cpu().numpy() - dec[t]).max() < TOL_F32_CHAIN


def test_chain_bf16_vs_f32_reported(W):
    """bf16 operands over a T=100 chain with identical noise: deviation is measured and bounded."""
    T, B, N = 100, 4, 256
    ef, eb = _engine(W, T, "f32"), _engine(W, T, "bf16")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=5)
    seg = torch.from_numpy(synth.make_seg_mask(valid, N))
    var = np.exp(logvar).astype(np.float32)
    rng = np.random.default_rng(10)
    xT = torch.from_numpy(rng.standard_normal((B, 3, N)).astype(np.float32))
    zs = torch.from_numpy(rng.standard_normal((T, B, 3, N)).astype(np.float32))
    args = tuple(map(torch.from_numpy, (part_code, mean, var, valid)))
    pf, _ = ef.sample_chain(ef.prepare_shapes(*args), seg, x_T_noise=xT, step_noise=zs)
    pb, _ = eb.sample_chain(eb.prepare_shapes(*args), seg, x_T_noise=xT, step_noise=zs)
    d = (pf - pb).abs()
    scale = float(np.sqrt(var).mean())
    print(f"bf16 vs f32 chain T=100: max-abs {d.max().item():.3e}, mean-abs {d.mean().item():.3e}, part sigma ~{scale:.3f}")
    assert d.max().item() < TOL_BF16_CHAIN_T100


def test_philox_chain_deterministic_and_statistical(W):
    T, B, N = 10, 8, 512
    eng = _engine(W, T, "f32")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=3, all_valid=True)
    seg = torch.from_numpy(synth.make_seg_mask(valid, N))
    var = np.exp(logvar).astype(np.float32)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    p1, t1 = eng.sample_chain(ctx, seg, seed=1234, ret_interval=10)
    p2, _ = eng.sample_chain(ctx, seg, seed=1234)
    p3, _ = eng.sample_chain(ctx, seg, seed=1235)
    assert torch.equal(p1, p2) and not torch.equal(p1, p3)
    # x_T = sqrt(var) z + anchors: per-part standardised prior sample is N(0,1)
    xT = t1[0].cpu().numpy()                                  # (B,N,3), t = T snapshot
    a, v = odf.gather_params(seg.numpy(), mean, var)
    z = (xT.transpose(0, 2, 1) - a) / np.sqrt(v)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3) < 0.15
    assert torch.isfinite(p1).all()


def test_argument_validation(eng10):
    g = np.load(os.path.join(GOLDEN, "denoiser_eps_B2_N128_mixed.npz"))
    ctx = _prep(eng10, g)
    # the C entry points take N % 32 == 0 and say so; the engine pads other sizes (test_point_counts_that_are_not_multiples_of_32)
    from difffacto_amd import _ffi
    x48, s48, o48 = torch.zeros(2, 3, 48, device="cuda"), torch.zeros(2, 48, dtype=torch.int32, device="cuda"), torch.empty(2, 3, 48, device="cuda")
    rc = _ffi.lib().dfx_denoise_eps(eng10._h, _ffi.ptr(ctx.buf), _ffi.ptr(x48), _ffi.ptr(s48), 0, _ffi.ptr(o48), 2, 48, _ffi.current_stream())
    with pytest.raises(RuntimeError, match="multiple of 32"):
        _ffi.check(rc, "dfx_denoise_eps")
    assert eng10.eps(ctx, x48, s48, 0).shape == (2, 3, 48)
    with pytest.raises(RuntimeError, match="outside"):
        eng10.eps(ctx, torch.from_numpy(g["x"]), torch.from_numpy(g["seg"]), 10)
    # empty batch is a no-op
    out = eng10.eps(ctx, torch.zeros(0, 3, 64), torch.zeros(0, 64, dtype=torch.int32), 0)
    assert out.shape == (0, 3, 64)


# ------------------------------------------------------------------------------------------------ DDIM
DDIM_CASES = {"quad8_eta1": dict(ddim_nsteps=8, ddim_discretize="quad", ddim_eta=1.0),
              "uniform5_eta0": dict(ddim_nsteps=5, ddim_discretize="uniform", ddim_eta=0.0)}


@pytest.mark.parametrize("name", sorted(DDIM_CASES))
def test_ddim_chain_f32_vs_reference_golden(W, name):
    """The reference's ddim_sampling branch (its own p_sample_loop_progressive / decode, T = 40) vs dfx_sample_chain_ddim
    (one launch) and dfx_p_sample_ddim (generator protocol); 'quad' visits t = 0 twice."""
    g = np.load(os.path.join(GOLDEN, f"ddim_T40_{name}_B2_N64.npz"))
    eng = _engine(W, 40, "f32")
    ctx = _prep(eng, g)
    seg = torch.from_numpy(g["seg"])
    steps, eta, ri = g["steps"].tolist(), float(g["ddim_eta"]), int(g["ret_interval"])
    pred, traj = eng.sample_chain_ddim(ctx, seg, steps, eta, x_T_noise=torch.from_numpy(g["x_T_noise"]),
                                       step_noise=torch.from_numpy(g["step_noise"]), ret_interval=ri)
    assert np.abs(pred.cpu().numpy() - g["decode_pred"]).max() < TOL_F32_CHAIN
    for k, t in enumerate(eng.snapshot_times(ri)):
        if f"decode_{t}" in g.files:
            assert np.abs(traj[k].cpu().numpy() - g[f"decode_{t}"]).max() < TOL_F32_CHAIN, t
    x = torch.from_numpy(g["traj"][0]).cuda()
    for i, t in enumerate(steps[::-1]):
        x = eng.p_sample_ddim(ctx, x, seg, t, eta, noise=torch.from_numpy(g["step_noise"][i]))
        assert np.abs(x.cpu().numpy() - g["traj"][i + 1]).max() < TOL_F32_CHAIN, (i, t)


def test_ddim_pipelined_bf16_vs_f32_and_module_api(W):
    """N = 256 takes the LDS-pipelined bf16 kernel: compare with the exact fp32 path on identical noise, through the
    AnchoredDiffusion mirror built with the reference's ddim arguments."""
    from difffacto_amd.modules import AnchoredDiffusion, decode
    from test_modules_cpu import DIFF_CFG
    B, N, T = 4, 256, 100
    out = {}
    part_code, mean, logvar, valid = synth.make_latents(B, seed=9)
    seg = torch.from_numpy(synth.make_seg_mask(valid, N)).cuda()
    ctx = [torch.from_numpy(part_code).cuda(), torch.from_numpy(np.concatenate([mean, np.exp(logvar)], 1).astype(np.float32)).cuda()]
    rng = np.random.Generator(np.random.PCG64(4))
    xT = torch.from_numpy(rng.standard_normal((B, 3, N)).astype(np.float32))
    for prec in ("f32", "bf16"):
        d = AnchoredDiffusion(num_timesteps=T, precision=prec, **{**DIFF_CFG, "ddim_sampling": True, "ddim_nsteps": 25,
                                                                  "ddim_discretize": "quad", "ddim_eta": 1.0})
        d.model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
        d = d.cuda().eval()
        sn = torch.from_numpy(rng.standard_normal((len(d.steps), B, 3, N)).astype(np.float32)) if prec == "f32" else sn
        out[prec] = decode(d, ctx, seg, valid_id=torch.from_numpy(valid).cuda(), ret_traj=True, ret_interval=10, x_T_noise=xT, step_noise=sn)
        assert sorted(k for k in out[prec] if k != "pred") == sorted({t for t in d.steps if t and t % 10 == 0} | {T})
    err = (out["f32"]["pred"] - out["bf16"]["pred"]).abs().max().item()
    print(f"ddim bf16 vs f32 (quad, 25 steps, T=100, B=4, N=256): max-abs {err:.3e}")
    assert torch.isfinite(out["bf16"]["pred"]).all() and err < 2.7e-4, err   # measured 8.8e-5 (gpurun r03): gate at 3x


# ---------------------------------------------------------------------- training-style forward (eval mode)
@pytest.mark.parametrize("prec,tol", [("f32", 1e-5), ("bf16", 5e-3)])
def test_training_losses_forward_vs_reference_golden(W, prec, tol):
    """q_sample + per-shape-t denoiser + masked MSE vs the reference's own training_losses (eval mode, dropout off)."""
    from difffacto_amd.modules import AnchoredDiffusion
    from test_modules_cpu import DIFF_CFG
    g = np.load(os.path.join(GOLDEN, "train_fwd_B3_N64_T10.npz"))
    d = AnchoredDiffusion(num_timesteps=10, precision=prec, **DIFF_CFG)
    d.model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    d = d.cuda().eval()
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ctx = [c(g["part_code"]), c(np.concatenate([g["mean"], np.exp(g["logvar"])], 1).astype(np.float32))]
    seg, valid = c(g["seg"]), c(g["valid"])
    eng = d.model.engine()
    sc = d._sc(ctx, valid)
    x_t = eng.q_sample(sc, seg, c(g["x_start"]), g["t"], c(g["noise"]))
    assert np.abs(x_t.cpu().numpy() - g["x_t"]).max() < 1e-6
    with torch.no_grad():
        for name, fl in (("flags", c(g["flags"])), ("noflags", None)):
            r = d.training_losses(c(g["x_start"]), c(g["t"]), ctx=ctx, anchor_assignment=seg, valid_id=valid, flags=fl, noise=c(g["noise"]))
            ref = float(g["mse_loss_" + name])
            assert abs(float(r["mse_loss"]) - ref) < tol * max(1.0, ref), (name, float(r["mse_loss"]), ref)
    with pytest.raises(NotImplementedError):
        d.training_losses(c(g["x_start"]), c(g["t"]), ctx=ctx, anchor_assignment=seg, valid_id=valid, noise=c(g["noise"]))   # grad mode without anchors / variance
    # the pipelined kernel (N % 256 == 0) with per-shape t agrees with the direct kernel
    B, N = 4, 256
    pc, mean, logvar, va = synth.make_latents(B, seed=3)
    e = _engine(W, 100, "bf16")
    cx = e.prepare_shapes(*map(torch.from_numpy, (pc, mean, np.exp(logvar).astype(np.float32), va)))
    sg = torch.from_numpy(synth.make_seg_mask(va, N))
    x = torch.randn(B, 3, N)
    tt = torch.tensor([0, 37, 99, 5])
    from difffacto_amd import _ffi
    a = e.eps_t(cx, x, sg, tt)
    _ffi.lib().dfx_debug_force_direct(1)
    b = e.eps_t(cx, x, sg, tt)
    _ffi.lib().dfx_debug_force_direct(0)
    one = torch.stack([e.eps(cx, x, sg, int(t))[i] for i, t in enumerate(tt)])
    print(f"pipelined vs direct bf16 kernel, one evaluation: max-abs {(a - b).abs().max().item():.3e}")
    assert (a - b).abs().max().item() < TOL_PIPE_VS_DIRECT and (a - one).abs().max().item() == 0.0   # `one` also takes the pipelined kernel


def test_pipelined_kernel_any_batch_size_matches_direct(W):
    """The XCD-aware workgroup order remaps whole groups of 8 shapes and leaves the remainder in natural order: every
    shape must come out the same through the pipelined and the direct bf16 kernels for B = 12 (8 remapped + 4 natural)
    and B = 3 (no remap), N = 2048, explicit noise."""
    from difffacto_amd import _ffi
    N, T = 2048, 3
    e = _engine(W, T, "bf16")
    for B in (12, 3):
        pc, mean, logvar, va = synth.make_latents(B, seed=B)
        cx = e.prepare_shapes(*map(torch.from_numpy, (pc, mean, np.exp(logvar).astype(np.float32), va)))
        sg = torch.from_numpy(synth.make_seg_mask(va, N))
        g = torch.Generator().manual_seed(B)
        xT, sn = torch.randn(B, 3, N, generator=g), torch.randn(T, B, 3, N, generator=g)
        a, _ = e.sample_chain(cx, sg, x_T_noise=xT, step_noise=sn)
        _ffi.lib().dfx_debug_force_direct(1)
        b, _ = e.sample_chain(cx, sg, x_T_noise=xT, step_noise=sn)
        _ffi.lib().dfx_debug_force_direct(0)
        per_shape = (a - b).abs().amax(dim=(1, 2))
        print(f"pipelined vs direct bf16 chain T={T} B={B}: max-abs per shape {per_shape.max().item():.3e}")
        assert per_shape.max().item() < TOL_PIPE_VS_DIRECT, per_shape   # same bf16 operands, different (fast) GELU / LayerNorm formulation


@pytest.mark.parametrize("N", [2048, 2080, 512])
def test_small_batch_workgroup_sizes_are_bit_identical(W, N):
    """The pipelined kernel runs 8, 4 or 2 wavefronts per workgroup (small batches take the smaller ones so that a single
    shape spreads over 16 / 32 CUs instead of 8), and the smallest batches take the co-operative latency kernel (`1`: one
    32-point tile per workgroup, eight wavefronts on it): same floating-point operations in the same order per accumulator, so
    the clouds must agree bit for bit, with explicit noise and with the in-kernel Philox stream, full and ragged last tiles,
    and so must single evaluations / single steps."""
    from difffacto_amd import _ffi
    T, B = 4, 3
    e = _engine(W, T, "bf16")
    pc, mean, logvar, va = synth.make_latents(B, seed=N)
    cx = e.prepare_shapes(*map(torch.from_numpy, (pc, mean, np.exp(logvar).astype(np.float32), va)))
    sg = torch.from_numpy(synth.make_seg_mask(va, N))
    g = torch.Generator().manual_seed(N)
    xT, sn = torch.randn(B, 3, N, generator=g), torch.randn(T, B, 3, N, generator=g)
    out = {}
    try:
        for nw in (8, 4, 2, 1, 64, 16):   # 64 = k_denoise_pipe2: two point tiles per wavefront, four wavefronts per workgroup; 16 = k_denoise_coop2: two tiles per co-operative workgroup
            _ffi.lib().dfx_debug_pipe_waves(nw)
            out[nw] = (e.sample_chain(cx, sg, x_T_noise=xT, step_noise=sn, ret_interval=2), e.sample_chain(cx, sg, seed=9)[0],
                       e.eps(cx, xT, sg, 2), e.p_sample(cx, xT, sg, 1, noise=sn[0], want_xstart=True))
    finally:
        _ffi.lib().dfx_debug_pipe_waves(0)
    auto = e.sample_chain(cx, sg, seed=9)[0]     # B = 3 is a small batch: the automatic choice is one of the three
    for nw in (4, 2, 1, 64, 16):
        assert torch.equal(out[nw][0][0], out[8][0][0]) and torch.equal(out[nw][0][1], out[8][0][1]), nw
        assert torch.equal(out[nw][1], out[8][1]), nw
        assert torch.equal(out[nw][2], out[8][2]), nw
        assert torch.equal(out[nw][3][0], out[8][3][0]) and torch.equal(out[nw][3][1], out[8][3][1]), nw
    assert torch.equal(auto, out[8][1]) and torch.isfinite(auto).all()


@pytest.mark.parametrize("N", [2048, 2080, 512, 96])
def test_f32_pipelined_chain_is_bit_identical_to_the_direct_kernel(W, N):
    """k_denoise_pipe_f32 (weights through the LDS ring, 8 / 4 / 2 wavefronts per workgroup) against the direct exact-fp32 kernel
    that every fp32 golden was established on: same device functions, same MFMA order per accumulator -> the same bits, for the
    chain with explicit noise (trajectory included) and with the in-kernel Philox stream, single evaluations (one t and per-shape t),
    single steps with pred_xstart, and the DDIM chain; full, ragged (2080) and padded (96: three tiles of 32 in a 64-point tile pair)
    last tiles."""
    from difffacto_amd import _ffi
    T, B = 4, 3
    e = _engine(W, T, "f32")
    pc, mean, logvar, va = synth.make_latents(B, seed=N + 1)
    cx = e.prepare_shapes(*map(torch.from_numpy, (pc, mean, np.exp(logvar).astype(np.float32), va)))
    sg = torch.from_numpy(synth.make_seg_mask(va, N))
    g = torch.Generator().manual_seed(N)
    xT, sn = torch.randn(B, 3, N, generator=g), torch.randn(T, B, 3, N, generator=g)
    tt = torch.tensor([3, 0, 2], dtype=torch.int32)

    def run():
        return (e.sample_chain(cx, sg, x_T_noise=xT, step_noise=sn, ret_interval=2), e.sample_chain(cx, sg, seed=9)[0], e.eps(cx, xT, sg, 2),
                e.p_sample(cx, xT, sg, 1, noise=sn[0], want_xstart=True), e.eps_t(cx, xT, sg, tt),
                e.sample_chain_ddim(cx, sg, [0, 1, 3], 1.0, x_T_noise=xT, step_noise=sn[:3])[0])
    out = {}
    try:
        _ffi.lib().dfx_debug_force_direct(1)
        out[0] = run()
        _ffi.lib().dfx_debug_force_direct(0)
        for nw in (8, 4, 2):
            _ffi.lib().dfx_debug_pipe_waves(nw)
            out[nw] = run()
    finally:
        _ffi.lib().dfx_debug_force_direct(0)
        _ffi.lib().dfx_debug_pipe_waves(0)
    out["auto"] = run()
    flat = lambda r: [r[0][0], r[0][1], r[1], r[2], r[3][0], r[3][1], r[4], r[5]]
    ref = flat(out[0])
    assert all(torch.isfinite(x).all() for x in ref)
    for k in (8, 4, 2, "auto"):
        for i, (x, y) in enumerate(zip(flat(out[k]), ref)):
            assert torch.equal(x, y), (k, i, (x - y).abs().max().item())


def _tile_golden(g, reps):
    return {k: np.concatenate([g[k]] * reps, axis=0) for k in ("part_code", "mean", "logvar", "valid", "x", "seg")}


def trained_like(W, seed=5):
    """Synthetic weights with the statistics trained LayerNorm gains tend to have: heavy-tailed gamma (log-normal, sigma 0.5) with a few
    outlier channels per LayerNorm (x 16 .. x 64, channel 127 of norm3 among them in two blocks) and non-zero beta."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Wt = {k: v.copy() for k, v in W.items()}
    for k in Wt:
        if ".norm" in k or k.startswith(("pre_norm", "post_norm")):
            if k.endswith(".weight"):
                Wt[k] = (Wt[k] * np.exp(0.5 * rng.standard_normal(128))).astype(np.float32)
                Wt[k][rng.choice(127, size=3, replace=False)] *= rng.choice([16.0, 32.0, 64.0], size=3).astype(np.float32)
            else:
                Wt[k] = (Wt[k] + 0.3 * rng.standard_normal(128)).astype(np.float32)
    Wt["transformer_blocks.1.norm3.weight"][127] *= 32.0
    Wt["transformer_blocks.3.norm3.weight"][127] *= 64.0
    return Wt


def test_w1_bias_fold_moves_to_another_channel_around_an_outlier_and_is_selectable(W):
    """The bf16 pack carries b1' in ONE hidden channel's K slot of W1 and subtracts that channel's column of W1 diag(gamma3) from every other
    column — exact in real arithmetic (a LayerNorm output sums to zero), but every weight of a row then rounds with a step that follows
    that column.  Round 5 backed off to the plain pack + the ~3x slower direct kernel when channel 127's column was an outlier
    (VERDICT r5 weak #8); round 6: dfx_denoiser_create exchanges 127 with the hidden channel whose column is the smallest over all blocks
    (a relabelling of the residual stream's channels: the same function), so the fold — and the pipelined / co-operative kernels — stay.
    Checked against the exact fp32 engine on the N = 2048 golden inputs (B = 1: co-operative kernel) and 64 copies of them (pipelined):
      * synthetic weights: folded on 127 by default (ratio ~1-3); forced-plain agrees with fp32 within the bf16 gate as well;
      * gamma3[127] of block 2 times 64: folded on ANOTHER channel, the fast kernels, error <= the plain pack's; forcing the fold onto 127
        (dfx_debug_w1_fold(1), round 5's form) on those weights is measurably worse;
      * heavy-tailed "trained-like" gains with several outlier channels: folded, fast kernels, within the plain pack's error."""
    from difffacto_amd import _ffi
    from difffacto_amd.engine import last_kernel_variant
    g1 = dict(np.load(os.path.join(GOLDEN, "denoiser_eps_B1_N2048.npz")))
    t = int(g1["ts"][0])

    def err(Wx, mode, reps=1):
        g = _tile_golden(g1, reps)
        x, seg = torch.from_numpy(g["x"]), torch.from_numpy(g["seg"])
        _ffi.lib().dfx_debug_w1_fold(mode)
        try:
            eb = _engine(Wx, 10, "bf16")
        finally:
            _ffi.lib().dfx_debug_w1_fold(-1)
        ef = _engine(Wx, 10, "f32")
        out = eb.eps(_prep(eb, g), x, seg, t).cpu().numpy()
        variant = last_kernel_variant()
        ref = ef.eps(_prep(ef, g), x, seg, t).cpu().numpy()
        assert all(np.array_equal(out[0], out[i]) for i in range(1, reps))
        return float(np.abs(out - ref).max()), eb.w1_fold(), eb.w1_fold_channel(), variant, float(np.abs(ref).max())

    e_def, (folded, ratio), ch, variant, scale = err(W, -1)
    assert folded and ch == 127 and ratio < 4 and variant != "k_denoise<bf16>", (folded, ch, ratio, variant)   # (B = 1: a co-operative chain kernel)
    e_plain, (folded_p, _), ch_p, variant_p, _ = err(W, 0)
    assert not folded_p and ch_p == -1 and variant_p in ("k_denoise<bf16>", "k_denoise_coop16"), variant_p
    assert e_def <= TOL_BF16_EPS and e_plain <= TOL_BF16_EPS, (e_def, e_plain)
    Wo = {k: v.copy() for k, v in W.items()}
    Wo["transformer_blocks.2.norm3.weight"][127] *= 64.0
    e_auto, (folded_o, ratio_o), ch_o, variant_o, scale_o = err(Wo, -1)
    assert folded_o and ch_o not in (127, -1) and ratio_o < 4 and variant_o not in ("k_denoise<bf16>",), (folded_o, ch_o, ratio_o, variant_o)
    e_auto64, _, _, variant_o64, _ = err(Wo, -1, reps=64)
    assert variant_o64 == "k_denoise_pipe<8>", variant_o64            # the headline kernel, not the direct one
    e_oplain, (folded_op, _), _, _, _ = err(Wo, 0)
    e_forced, (folded_f, ratio_f), ch_f, _, _ = err(Wo, 1)
    assert folded_f and ch_f == 127 and ratio_f > 8 and not folded_op
    Wt = trained_like(W)
    e_t, (folded_t, ratio_t), ch_t, variant_t, scale_t = err(Wt, -1, reps=64)
    e_tplain, _, _, _, _ = err(Wt, 0)
    assert folded_t and ch_t not in (127, -1) and variant_t == "k_denoise_pipe<8>", (folded_t, ch_t, ratio_t, variant_t)
    print(f"W1 bias fold: synthetic weights (channel {ch}, ratio {ratio:.2f}): folded {e_def:.2e} / plain {e_plain:.2e} (|eps| {scale:.2f}); "
          f"gamma3[127] x 64: moved to channel {ch_o} (ratio {ratio_o:.2f}) {e_auto:.2e} [{variant_o}], x64 shapes {e_auto64:.2e} [{variant_o64}], "
          f"plain {e_oplain:.2e}, forced onto 127 (ratio {ratio_f:.1f}) {e_forced:.2e} (|eps| {scale_o:.2f}); "
          f"trained-like gains: channel {ch_t} (ratio {ratio_t:.2f}) {e_t:.2e} [{variant_t}] / plain {e_tplain:.2e} (|eps| {scale_t:.2f})")
    # measured (r05): synthetic weights folded 3.5e-3 / plain 2.0e-3; outlier weights plain 7.7e-3 (the outlier channel's own bf16 rounding, in any
    # bf16 formulation) / forced fold on 127 1.4e-1.  The moved fold must stay within 1.5x of the plain pack on the same weights (and the 2x gate).
    assert e_auto <= 2 * TOL_BF16_EPS and e_auto <= 1.5 * e_oplain + 1e-3 and e_auto64 <= 2 * TOL_BF16_EPS, (e_auto, e_oplain, e_auto64)
    assert e_forced > 5 * e_auto, (e_forced, e_auto)
    assert e_t <= 1.5 * e_tplain + 1e-3 * scale_t, (e_t, e_tplain, scale_t)

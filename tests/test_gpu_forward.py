"""GPU parity of the top-level compositions against goldens produced by the reference's own classes (tests/golden/make_golden_forward.py):

* ``AnchorDiffAE.forward`` — the gen branch (anchor_gen.py:1034-1084) and the encode -> decode "sample" mode (:1085-1134): every key of
  the dict ``Runner.val`` saves (runner/runner.py:372-377), with the reference's random draws replayed at the same sites;
* ``PartEncoder.forward`` with the part aligner + ``sample_noise`` (part_encoders.py:1185-1260, :388-414);
* the seeding contract of the module API: seed-less calls draw FRESH noise per call and replay under ``torch.manual_seed``.

Tolerances: integer / pass-through keys bit-exact; fp32 engine 2e-4 x max(1, |ref|) (a 10-step chain after a 14-layer flow and the
aligner: summation-order differences only); bf16 engine 3e-2 (operand rounding through the chain; stated, not assumed: measured ~4e-3).
"""
import os

import numpy as np
import pytest
import torch

from difffacto_amd import synth
from _replay import load_forward_fixture, replay_draws

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _model(T, N, K, gen, precision, ret_interval=5):
    from difffacto_amd.networks import AnchorDiffAE
    from test_modules_cpu import model_cfg
    # every key of the unmodified configs/gen_chair.py `cfg.model` (pinned against the reference's config loader on CPU:
    # test_install_full_builds_the_mirror_from_the_unmodified_configs), with the fixture's T / N / K / ret_interval
    m = AnchorDiffAE(**model_cfg(num_timesteps=T, npoints=N, gen=gen, cimle_sample_num=K, ret_interval=ret_interval), precision=precision)
    W = {"diffusion.model." + k: v for k, v in synth.make_denoiser_weights(0).items()}
    W.update({"encoder." + k: v for k, v in synth.make_latent_weights(0).items()})
    W.update({"encoder.encoder." + k: v for k, v in synth.make_pointnet_v2_weights(0).items()})
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return m.cuda().eval()


def _inject_chain_noise(model, x_T, steps):
    orig = model.decode

    def decode(*a, **k):
        return orig(*a, x_T_noise=torch.from_numpy(x_T).cuda(), step_noise=torch.from_numpy(steps).cuda(), **k)

    model.decode = decode


def _compare(pred, expect, tol):
    assert set(map(str, pred)) == set(expect), (sorted(map(str, pred)), sorted(expect))
    worst = 0.0
    for k, v in pred.items():
        ref = expect[str(k)]
        got = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind in "iub":
            assert np.array_equal(got, ref), k
        else:
            err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
            assert err <= tol, (k, err)
            worst = max(worst, err)
    return worst


@pytest.mark.parametrize("prec,tol", [("f32", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("tag,chain_at", [("gen_B2_K2_T10", 3), ("sample_B2_K2_T10", 4)])
def test_anchor_diff_ae_forward_matches_reference(tag, chain_at, prec, tol):
    batch, draws, expect, meta = load_forward_fixture(os.path.join(GOLDEN, f"forward_{tag}.npz"))
    T, K = int(meta["T"]), int(meta["K"])
    N = batch["ref"].shape[1]
    model = _model(T, N, K, gen=tag.startswith("gen"), precision=prec, ret_interval=int(meta["ret_interval"]))
    chain = draws[chain_at:chain_at + T + 1]                 # x_T (anchored_diffusion.py:564), then one randn_like per step (:476)
    rest = draws[:chain_at] + draws[chain_at + T + 1:]
    _inject_chain_noise(model, chain[0], np.stack(chain[1:]))
    with replay_draws(rest) as queue:
        out = model(batch, device="cuda", epoch=0)
    assert not queue, f"{len(queue)} recorded draws were not consumed"
    assert len(out) == 1 and out[0][1] == str(meta["name"])
    pred = out[0][0]
    assert all((not isinstance(v, torch.Tensor)) or v.device.type == "cpu" for v in pred.values())   # anchor_gen.py:1082 / :1129
    worst = _compare(pred, expect, tol)
    print(f"forward[{tag}, {prec}]: {len(pred)} keys, worst rel err {worst:.2e}")


def test_encoder_forward_with_part_aligner_matches_reference():
    g = np.load(os.path.join(GOLDEN, "encoder_fwd_B3_N96.npz"))
    batch, draws, expect, meta = load_forward_fixture(os.path.join(GOLDEN, "encoder_fwd_B3_N96.npz"))
    model = _model(10, 96, 1, True, "f32")
    enc = model.encoder
    enc.kl_weight = float(meta["kl_weight"])
    num, epoch = int(meta["num"]), int(meta["epoch"])
    with replay_draws(draws) as queue, torch.no_grad():
        ctx, mpp, lpp, fpp, losses, lat = enc(batch, "cuda", epoch=epoch)
        noise, idx = enc.sample_noise(batch, "cuda", num)
        ctx2, mpp2, lpp2, fpp2, losses2, lat2 = enc(batch, "cuda", noise=noise, epoch=epoch)
    assert not queue
    got = {"ctx0": ctx[0], "ctx1": ctx[1], "mean_pp": mpp, "logvar_pp": lpp, "flag_pp": fpp, "part_code": lat[0], "mean": lat[1],
           "logvar": lat[2], "noise": lat[3], "sn_noise": noise, "sn_id": idx, "k_ctx0": ctx2[0], "k_ctx1": ctx2[1], "k_mean_pp": mpp2,
           "k_logvar_pp": lpp2, "k_flag_pp": fpp2, "k_fit_loss": losses2["fit_loss"]}
    got.update({"loss/" + k: v for k, v in losses.items()})
    got = {k: v.detach().cpu().reshape(expect[k].shape) if isinstance(v, torch.Tensor) else v for k, v in got.items()}
    worst = _compare(got, expect, 2e-4)
    print(f"encoder forward + sample_noise: worst rel err {worst:.2e}; arg-min ids {idx.tolist()}")
    # AnchorDiffAE.cache_noise (anchor_gen.py:807-815, the cIMLE noise cache of stage 2) = the arg-min candidate of sample_noise: same draws -> same rows
    model.sample_noise_num = num
    with replay_draws(draws[1:3]) as queue:
        cached = model.cache_noise(batch, "cuda")
    assert not queue
    ref = expect["sn_noise"][np.arange(expect["sn_noise"].shape[0]), expect["sn_id"]]
    assert np.array_equal(cached.cpu().numpy(), ref)


def test_seedless_sampling_is_fresh_per_call_and_replays_under_manual_seed():
    """The reference's decode draws x_T and every z_t from the global generator (anchored_diffusion.py:476,564): two consecutive calls
    differ, and torch.manual_seed replays them.  Same contract here for decode / sample_chain / p_sample / the generator loop."""
    from difffacto_amd.modules import decode
    model = _model(10, 64, 1, True, "bf16")
    d = model.diffusion
    pc, mean, logvar, valid = synth.make_latents(3, seed=5)
    cu = lambda a: torch.from_numpy(a).cuda()
    ctx = [cu(pc), torch.cat([cu(mean), torch.exp(cu(logvar))], 1)]
    seg = cu(synth.make_seg_mask(valid, 64))
    torch.manual_seed(1234)
    a = decode(d, ctx, seg, cu(valid))["pred"]
    b = decode(d, ctx, seg, cu(valid))["pred"]
    assert not torch.equal(a, b), "two seed-less decode() calls replayed the same noise"
    assert float((a - b).abs().mean()) > 1e-2
    torch.manual_seed(1234)
    a2 = decode(d, ctx, seg, cu(valid))["pred"]
    b2 = decode(d, ctx, seg, cu(valid))["pred"]
    assert torch.equal(a, a2) and torch.equal(b, b2)
    # an explicit torch.Generator governs the call instead of the global one
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    c1 = decode(d, ctx, seg, cu(valid), generator=g1)["pred"]
    c2 = decode(d, ctx, seg, cu(valid), generator=g2)["pred"]
    assert torch.equal(c1, c2) and not torch.equal(c1, a)
    # the reference-protocol generator (what AnchorDiffAE.decode of the REFERENCE consumes) and single steps
    idx = seg.long()[:, None, :].expand(-1, 3, -1)
    anchors, variance = torch.gather(ctx[1][:, :3], 2, idx), torch.gather(ctx[1][:, 3:], 2, idx)

    def loop():
        last = None
        for t, s in d.p_sample_loop_progressive([3, 3, 64], anchors, ctx=ctx, variance=variance, anchor_assignment=seg, valid_id=cu(valid)):
            last = s["sample"]
        return last
    torch.manual_seed(99)
    l1, l2 = loop(), loop()
    torch.manual_seed(99)
    l3 = loop()
    assert not torch.equal(l1, l2) and torch.equal(l1, l3)
    x = torch.sqrt(variance) * torch.randn(3, 3, 64, device="cuda") + anchors
    s1 = d.p_sample(x, 5, anchors, ctx=ctx, variance=variance, anchor_assignment=seg, valid_id=cu(valid))["sample"]
    s2 = d.p_sample(x, 5, anchors, ctx=ctx, variance=variance, anchor_assignment=seg, valid_id=cu(valid))["sample"]
    s3 = d.p_sample(x, 5, anchors, ctx=ctx, variance=variance, anchor_assignment=seg, valid_id=cu(valid), seed=3)["sample"]
    s4 = d.p_sample(x, 5, anchors, ctx=ctx, variance=variance, anchor_assignment=seg, valid_id=cu(valid), seed=3)["sample"]
    assert not torch.equal(s1, s2) and torch.equal(s3, s4)


def test_decode_with_pred_xstart_outputs():
    """anchor_gen.py:160-167 (save_pred_xstart): 'pred_xstart' / 'pred_xstart_{t}' next to the samples; with explicit noise the samples are
    the single-launch chain's, and pred_xstart of the last step equals its sample (posterior_mean_coef2/3 = 0 and coef1 = 1 at t = 0)."""
    from difffacto_amd.modules import decode
    model = _model(10, 64, 1, True, "f32")
    d = model.diffusion
    pc, mean, logvar, valid = synth.make_latents(2, seed=6)
    cu = lambda a: torch.from_numpy(a).cuda()
    ctx = [cu(pc), torch.cat([cu(mean), torch.exp(cu(logvar))], 1)]
    seg = cu(synth.make_seg_mask(valid, 64))
    rng = np.random.default_rng(0)
    xT, zs = cu(rng.standard_normal((2, 3, 64)).astype(np.float32)), cu(rng.standard_normal((10, 2, 3, 64)).astype(np.float32))
    ref = decode(d, ctx, seg, cu(valid), ret_traj=True, ret_interval=5, x_T_noise=xT, step_noise=zs)
    out = decode(d, ctx, seg, cu(valid), ret_traj=True, ret_interval=5, x_T_noise=xT, step_noise=zs, save_pred_xstart=True)
    assert set(map(str, out)) == {"pred", "pred_xstart", "5", "pred_xstart_5", "10"}
    for k in ("pred", 5, 10):
        assert float((out[k] - ref[k]).abs().max()) <= 1e-5, k
    assert float((out["pred_xstart"] - out["pred"]).abs().max()) <= 1e-5
    assert torch.isfinite(out["pred_xstart_5"]).all() and tuple(out["pred_xstart_5"].shape) == (2, 64, 3)

"""GPU parity of the latent sampler through the C-ABI: dfx_flow_reverse, dfx_part_aligner,
dfx_sample_latents vs the reference goldens (tests/golden/latents_*.npz, produced by the reference's own
PartEncoder.sample_latents) and vs the numpy oracle on seeded inputs at the shipped batch size.

Tolerance: fp32 MFMA vs fp32 torch/numpy differ only by summation order; the 14-layer flow divides by
sigmoid scales 14 times, so the gate is 1e-4 x max(1, |ref|_inf); integer outputs (seg ids, validity) bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from difffacto_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _close(a, ref, tol=TOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == ref.shape, (a.shape, ref.shape)
    err = float(np.abs(a - ref).max())
    assert err <= tol * max(1.0, float(np.abs(ref).max())), err


@pytest.fixture(scope="module")
def sampler():
    from difffacto_amd.latents import LatentSampler
    return LatentSampler(synth.make_latent_weights(seed=0), noise_scale=100.0)


@pytest.mark.parametrize("tag", ["S3_K2_mixed", "S4_K3_fixed"])
def test_sample_latents_matches_reference_golden(sampler, tag):
    g = np.load(os.path.join(GOLDEN, f"latents_{tag}.npz"))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _close(sampler.flow_reverse(cu(g["w_noise"]))[..., 2], g["flow2_reverse"])
    S = g["w_noise"].shape[0]
    m, lv = sampler.part_aligner(cu(g["w_noise"]), cu(g["valid_in"]), cu(g["aligner_noise"][:S]))
    _close(m, g["aligner_mean"])
    _close(lv, g["aligner_logvar"])
    out = sampler.sample_latents(cu(g["w_noise"]), cu(g["aligner_noise"]), cu(g["valid_in"]), fixed_id=g["fixed_id"],
                                 K=int(g["K"]), npoints=int(g["npoints"]))
    assert np.array_equal(out["seg_mask"].cpu().numpy(), g["seg_mask"])
    assert np.array_equal(out["valid_id"].cpu().numpy(), g["valid_id"])
    for k in ("part_code", "mean", "logvar", "noise", "mean_per_point", "logvar_per_point"):
        _close(out[k], g[k])
    _close(out["part_code"], g["ctx0"])
    _close(out["params"], g["ctx1"])


def test_sample_latents_vs_oracle_full_batch(sampler):
    """Shipped val shape: 128 shapes x K=10 aligner noises, 2048 points (anchor_gen.py:1042)."""
    from oracle import latents as ol
    S, K, N = 128, 10, 2048
    W = synth.make_latent_weights(seed=0)
    rng = np.random.Generator(np.random.PCG64(5))
    w = rng.standard_normal((S, 256, 4)).astype(np.float32)
    an = rng.standard_normal((S * K, 32)).astype(np.float32)
    _, _, _, valid = synth.make_latents(S, seed=5)
    ref = ol.sample_latents(W, w, an, valid, [0, 0, 0, 0], K, N, noise_scale=100.0)
    cu = lambda a: torch.from_numpy(a).cuda()
    out = sampler.sample_latents(cu(w), cu(an), cu(valid), K=K, npoints=N)
    assert np.array_equal(out["seg_mask"].cpu().numpy(), ref["seg_mask"])
    for k in ("part_code", "mean", "logvar", "mean_per_point", "logvar_per_point"):
        _close(out[k], ref[k])
    _close(out["params"], ref["ctx"][1])


def test_latents_of_a_shard_match_the_full_batch_bit_for_bit_with_one_k_grouping(sampler):
    """The few-row fp32 products of the front end split their K sum over four wavefronts when a call has few tiles (mfma_linear.h): the grouping, and
    with it the last bits, follows the batch size.  dfx_debug_lin_split_k(1) fixes ONE grouping for every batch size: the latents of shapes 32..47
    computed alone (a rank's shard) are then the bits of the same shapes inside the batch of 128; integer outputs always are."""
    from difffacto_amd import _ffi
    S, K, N = 128, 2, 256
    rng = np.random.Generator(np.random.PCG64(41))
    w = rng.standard_normal((S, 256, 4)).astype(np.float32)
    an = rng.standard_normal((S * K, 32)).astype(np.float32)
    _, _, _, valid = synth.make_latents(S, seed=41)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    lo, hi = 32, 48
    _ffi.lib().dfx_debug_lin_split_k(1)
    try:
        full = sampler.sample_latents(cu(w), cu(an), cu(valid), K=K, npoints=N)
        part = sampler.sample_latents(cu(w[lo:hi]), cu(an[lo * K:hi * K]), cu(valid[lo:hi]), K=K, npoints=N)
    finally:
        _ffi.lib().dfx_debug_lin_split_k(-1)
    for k in ("part_code", "mean", "logvar", "mean_per_point", "logvar_per_point", "params", "seg_mask"):
        a, b = full[k].cpu().numpy()[lo * K:hi * K], part[k].cpu().numpy()
        assert a.shape == b.shape and np.array_equal(a, b), (k, float(np.abs(a.astype(np.float64) - b).max()))


def test_given_part_code_and_ragged_sizes(sampler):
    """part_code given (flows skipped, part_encoders.py:1053), S=1 / S=33 (partial 32-row tiles), K=1."""
    from oracle import latents as ol
    W = synth.make_latent_weights(seed=0)
    for S in (1, 33):
        rng = np.random.Generator(np.random.PCG64(S))
        code = rng.standard_normal((S, 256, 4)).astype(np.float32)
        an = rng.standard_normal((S, 32)).astype(np.float32)
        _, _, _, valid = synth.make_latents(S, seed=S)
        ref = ol.sample_latents(W, code, an, valid, [0, 0, 1, 0], 1, 64, noise_scale=100.0, part_code=code)
        cu = lambda a: torch.from_numpy(a).cuda()
        out = sampler.sample_latents(None, cu(an), cu(valid), fixed_id=[0, 0, 1, 0], K=1, npoints=64, part_code=cu(code))
        assert np.array_equal(out["seg_mask"].cpu().numpy(), ref["seg_mask"])
        assert np.array_equal(out["valid_id"].cpu().numpy(), ref["valid_id"])
        for k in ("part_code", "mean", "logvar"):
            _close(out[k], ref[k])


def test_bad_arguments_raise(sampler):
    with pytest.raises(RuntimeError):
        sampler.sample_latents(None, torch.zeros(2, 32).cuda(), torch.ones(2, 4).cuda(), K=1, npoints=64)   # neither w nor code
    with pytest.raises(RuntimeError):
        sampler.sample_latents(torch.zeros(2, 256, 4).cuda(), torch.zeros(2, 32).cuda(), torch.ones(2, 4).cuda(), K=1, npoints=66)


# ---------------------------------------------------------------------------------- module mirror (encoders.py)
def _mirror():
    from difffacto_amd.encoders import PartEncoderForTransformerDecoder
    from test_modules_cpu import ENC_CFG
    enc = PartEncoderForTransformerDecoder(**ENC_CFG)
    W = synth.make_latent_weights(0)
    W.update({"encoder." + k: v for k, v in synth.make_pointnet_v2_weights(0).items()})
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)   # only num_batches_tracked is missing
    return enc.cuda().eval()


@pytest.mark.parametrize("tag", ["S3_K2_mixed", "S4_K3_fixed"])
def test_encoder_mirror_sample_latents_replays_reference(tag, monkeypatch):
    """Drive the mirror exactly like AnchorDiffAE.sample drives the reference encoder, with torch.randn replaying the
    reference's recorded draws (same order: part codes :1054, then aligner noise :1065)."""
    g = np.load(os.path.join(GOLDEN, f"latents_{tag}.npz"))
    enc = _mirror()
    queue = [g["w_noise"], g["aligner_noise"]]

    def fake_randn(*shape, device=None, **kw):
        a = queue.pop(0)
        assert tuple(shape) == a.shape
        return torch.from_numpy(a).to(device)

    monkeypatch.setattr(torch, "randn", fake_randn)
    S = g["w_noise"].shape[0]
    ctx, mpp, lpp, seg, vid, (code, mean, logvar, noise) = enc.sample_latents(
        S, int(g["npoints"]), "cuda", fixed_id=torch.from_numpy(g["fixed_id"]), valid_id=torch.from_numpy(g["valid_in"]).cuda(),
        epoch=0, K=int(g["K"]))
    assert not queue
    assert seg.dtype == torch.int32 and np.array_equal(seg.cpu().numpy(), g["seg_mask"])
    assert np.array_equal(vid.cpu().numpy(), g["valid_id"])
    for a, k in ((ctx[0], "ctx0"), (ctx[1], "ctx1"), (mpp, "mean_per_point"), (lpp, "logvar_per_point"), (code, "part_code"),
                 (mean, "mean"), (logvar, "logvar"), (noise, "noise")):
        _close(a, g[k])
    # the aligner on its own, reference call signature
    m, lv = enc.part_aligner(torch.from_numpy(g["w_noise"]).cuda(), torch.from_numpy(g["valid_in"]).cuda(),
                             noise=torch.from_numpy(g["aligner_noise"][:S]).cuda())
    _close(m, g["aligner_mean"])
    _close(lv, g["aligner_logvar"])


def test_generate_end_to_end():
    """anchor_gen.py:1034-1084: latents -> fused chain.  Clouds are finite and each part's points sit around its anchor."""
    from difffacto_amd.encoders import generate
    from difffacto_amd.modules import AnchoredDiffusion
    from test_modules_cpu import DIFF_CFG
    d = AnchoredDiffusion(num_timesteps=10, precision="bf16", **DIFF_CFG)
    d.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()})
    d = d.cuda().eval()
    enc = _mirror()
    _, _, _, valid = synth.make_latents(3, seed=3)
    torch.manual_seed(0)
    out = generate(enc, d, 3, 256, valid_id=torch.from_numpy(valid).cuda(), fixed_id=[0, 0, 0, 0], K=2, seed=1)
    assert tuple(out["pred"].shape) == (6, 256, 3) and torch.isfinite(out["pred"]).all()
    assert tuple(out["pred_seg_mask"].shape) == (6, 256) and tuple(out["anchors"].shape) == (6, 256, 3)
    assert np.array_equal(out["present"].cpu().numpy(), np.repeat(valid, 2, axis=0))


def test_pointnet_v2_matches_reference_golden():
    """PointNetV2.forward (pointnet.py:187-213, eval): native kernels vs the reference class's output, and vs the numpy
    oracle (pinned to the same golden) at the shipped size (B = 16, N = 2048)."""
    from oracle import pointnet_v2 as opv
    g = np.load(os.path.join(GOLDEN, "pointnet_v2_B3_N200.npz"))
    enc = _mirror()
    x, attn = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["attn"]).cuda()
    with torch.no_grad():
        m, v = enc.get_part_code(x, attn)
    _close(m, g["m"], 2e-5)
    _close(v, g["v"], 2e-5)
    rng = np.random.Generator(np.random.PCG64(3))
    xn = rng.uniform(-1, 1, size=(16, 2048, 3)).astype(np.float32)
    an = np.eye(4, dtype=np.float32)[rng.integers(0, 4, size=(16, 2048))]
    with torch.no_grad():
        m, v = enc.encoder(torch.from_numpy(xn).cuda(), torch.from_numpy(an).cuda())
    mo, vo = opv.forward(synth.make_pointnet_v2_weights(0), xn, an)
    _close(m, mo, 2e-5)
    _close(v, vo, 2e-5)


def test_pointnet_v2_has_no_second_implementation():
    """Every call libdfx does not run natively raises (no CPU / PyTorch fallback): CPU tensors -> the reference's
    'CPU not supported'; eval with autograd, train() with B = 1 -> NotImplementedError."""
    enc = _mirror().encoder
    x, attn = torch.zeros(2, 64, 3), torch.ones(2, 64, 4)
    with torch.no_grad(), pytest.raises(RuntimeError, match="CPU not supported"):
        enc(x, attn)
    with torch.enable_grad(), pytest.raises(RuntimeError, match="CPU not supported"):
        enc(x, attn)
    with torch.enable_grad(), pytest.raises(NotImplementedError, match="eval"):
        enc(x.cuda(), attn.cuda())             # parameters require grad: a gradient could be asked for
    # a FROZEN encoder called outside no_grad runs natively, like the reference's nn.Module
    with torch.no_grad():
        m0, v0 = enc(x.cuda(), attn.cuda())
    enc.requires_grad_(False)
    with torch.enable_grad():
        m1, v1 = enc(x.cuda(), attn.cuda())
        assert torch.equal(m0, m1) and torch.equal(v0, v1) and not m1.requires_grad
        with pytest.raises(NotImplementedError, match="eval"):
            enc(x.cuda().requires_grad_(True), attn.cuda())
    enc.requires_grad_(True)
    enc.train()
    with pytest.raises(RuntimeError, match="CPU not supported"):
        enc(x, attn)
    with pytest.raises(NotImplementedError, match="batch >= 2"):
        enc(x[:1].cuda(), attn[:1].cuda())
    m, v = enc(x.cuda(), attn.cuda())          # the native training path
    assert m.grad_fn is not None and tuple(m.shape) == (2, 4, 256)

"""Pin the numpy oracle against golden vectors produced by the reference's own Python model
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from difffacto_amd import synth
from oracle import denoiser as dn
from oracle import diffusion as df

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL_EPS = 2e-5     # fp32 restatement vs reference, |eps| ~ 1
TOL_CHAIN = 1e-4   # BASELINE.md config 1 gate for the restatement


def _ctx(g):
    return [g["part_code"], np.concatenate([g["mean"], np.exp(g["logvar"])], axis=1).astype(np.float32)]


def _per_point(g):
    return df.gather_params(g["seg"], g["mean"], np.exp(g["logvar"]).astype(np.float32))


@pytest.fixture(scope="module")
def W():
    return synth.make_denoiser_weights(seed=0)


@pytest.mark.parametrize("tag", ["B2_N128_mixed", "B2_N128_allvalid", "B1_N2048"])
def test_denoiser_eps_matches_reference(W, tag):
    g = np.load(os.path.join(GOLDEN, f"denoiser_eps_{tag}.npz"))
    anchors, variance = _per_point(g)
    B = g["x"].shape[0]
    for t in g["ts"]:
        eps = dn.transformer_net_forward(W, g["x"], np.full((B,), t), _ctx(g), anchors.transpose(0, 2, 1),
                                         variance.transpose(0, 2, 1), g["valid"], g["seg"])
        ref = g[f"eps_t{int(t)}"]
        assert eps.shape == ref.shape
        assert np.abs(eps - ref).max() < TOL_EPS, (tag, t, np.abs(eps - ref).max())


@pytest.mark.parametrize("T", [10, 100, 1000])
def test_tables_match_reference(T):
    g = np.load(os.path.join(GOLDEN, f"tables_T{T}.npz"))
    tb = df.Tables(T)
    for name in g.files:
        mine = getattr(tb, name).astype(np.float32)
        assert np.array_equal(mine, g[name]), name


@pytest.mark.parametrize("tag", ["B2_N128_mixed", "B3_N64_allvalid"])
def test_chain_matches_reference(W, tag):
    g = np.load(os.path.join(GOLDEN, f"chain_T10_{tag}.npz"))
    T = 10
    tb = df.Tables(T)
    anchors, variance = _per_point(g)
    traj = []
    for t, out in df.p_sample_loop_progressive(tb, W, anchors, _ctx(g), variance, g["seg"], g["valid"],
                                               g["x_T_noise"], g["step_noise"]):
        traj.append(out["sample"])
    traj = np.stack(traj)
    assert traj.shape == g["traj"].shape
    err = np.abs(traj - g["traj"]).reshape(T + 1, -1).max(axis=1)
    assert err.max() < TOL_CHAIN, err
    dec = df.decode(tb, W, anchors, _ctx(g), variance, g["seg"], g["valid"], g["x_T_noise"], g["step_noise"],
                    ret_traj=True, ret_interval=int(g["ret_interval"]))
    keys = sorted(k for k in g.files if k.startswith("decode_"))
    assert sorted("decode_" + str(k) for k in dec) == keys
    for k, v in dec.items():
        assert np.abs(v - g["decode_" + str(k)]).max() < TOL_CHAIN


def test_seg_mask_relabels_absent_part():
    valid = np.array([[0, 1, 1, 0], [1, 1, 1, 1]], dtype=np.float32)
    seg = synth.make_seg_mask(valid, 16)
    assert seg.shape == (2, 16)
    assert seg[0].tolist() == [1] * 4 + [1] * 4 + [2] * 4 + [1] * 4
    assert seg[1].tolist() == [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4


# --------------------------------------------------------------------------- latent sampler
@pytest.mark.parametrize("tag", ["S3_K2_mixed", "S4_K3_fixed"])
def test_oracle_sample_latents_matches_reference(tag):
    from difffacto_amd import synth
    from oracle import latents as ol
    g = np.load(os.path.join(GOLDEN, f"latents_{tag}.npz"))
    W = synth.make_latent_weights(seed=int(g["weight_seed"]))
    # pieces on their own
    f2 = ol.flow_reverse(np.ascontiguousarray(g["w_noise"][..., 2]), W, 2, ol.flow_depth(W))
    assert np.abs(f2 - g["flow2_reverse"]).max() < 1e-4 * max(1.0, np.abs(g["flow2_reverse"]).max())
    S = g["w_noise"].shape[0]
    m, lv = ol.part_aligner_forward(W, g["w_noise"], g["valid_in"], g["aligner_noise"][:S])
    assert np.abs(m - g["aligner_mean"]).max() < 1e-4 and np.abs(lv - g["aligner_logvar"]).max() < 1e-4
    # whole sample_latents
    out = ol.sample_latents(W, g["w_noise"], g["aligner_noise"], g["valid_in"], g["fixed_id"], int(g["K"]), int(g["npoints"]))
    assert np.array_equal(out["seg_mask"], g["seg_mask"])
    assert np.array_equal(out["valid_id"], g["valid_id"])
    for k, ref in (("part_code", g["part_code"]), ("mean", g["mean"]), ("logvar", g["logvar"]),
                   ("mean_per_point", g["mean_per_point"]), ("logvar_per_point", g["logvar_per_point"]),
                   ("noise", g["noise"])):
        assert out[k].shape == ref.shape, k
        assert np.abs(out[k] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), k
    assert np.abs(out["ctx"][0] - g["ctx0"]).max() < 1e-3 and np.abs(out["ctx"][1] - g["ctx1"]).max() < 1e-4


def test_oracle_pointnet_v2_matches_reference():
    from oracle import pointnet_v2 as opv
    g = np.load(os.path.join(GOLDEN, "pointnet_v2_B3_N200.npz"))
    W = synth.make_pointnet_v2_weights(seed=int(g["weight_seed"]))
    m, v = opv.forward(W, g["x"], g["attn"])
    assert m.shape == g["m"].shape == (3, 4, 256)
    assert np.abs(m - g["m"]).max() < 2e-5 and np.abs(v - g["v"]).max() < 2e-5


DDIM_CASES = {"quad8_eta1": dict(ddim_nsteps=8, ddim_discretize="quad", ddim_eta=1.0),
              "uniform5_eta0": dict(ddim_nsteps=5, ddim_discretize="uniform", ddim_eta=0.0)}


@pytest.mark.parametrize("name", sorted(DDIM_CASES))
def test_ddim_chain_matches_reference(W, name):
    """the ddim_sampling branch (anchored_diffusion.py:114-124, :368-377, :480-481), T = 40."""
    g = np.load(os.path.join(GOLDEN, f"ddim_T40_{name}_B2_N64.npz"))
    tb = df.Tables(40, ddim_sampling=True, **DDIM_CASES[name])
    assert tb.steps == g["steps"].tolist()
    anchors, variance = _per_point(g)
    traj = np.stack([out["sample"] for _, out in df.p_sample_loop_progressive(
        tb, W, anchors, _ctx(g), variance, g["seg"], g["valid"], g["x_T_noise"], g["step_noise"])])
    assert traj.shape == g["traj"].shape
    assert np.abs(traj - g["traj"]).max() < TOL_CHAIN
    dec = df.decode(tb, W, anchors, _ctx(g), variance, g["seg"], g["valid"], g["x_T_noise"], g["step_noise"],
                    ret_traj=True, ret_interval=int(g["ret_interval"]))
    assert sorted("decode_" + str(k) for k in dec) == sorted(k for k in g.files if k.startswith("decode_"))
    for k, v in dec.items():
        assert np.abs(v - g["decode_" + str(k)]).max() < TOL_CHAIN


def test_training_losses_forward_matches_reference(W):
    """SURVEY §8 A18 (forward, eval mode): q_sample + per-shape-t denoiser + masked MSE."""
    g = np.load(os.path.join(GOLDEN, "train_fwd_B3_N64_T10.npz"))
    tb = df.Tables(10)
    anchors, variance = _per_point(g)
    for name, fl in (("flags", g["flags"]), ("noflags", None)):
        r = df.training_losses(tb, W, g["x_start"], g["t"], anchors, variance, _ctx(g), g["seg"], g["valid"], fl, g["noise"])
        assert np.abs(r["x_t"] - g["x_t"]).max() < 1e-6
        assert abs(float(r["mse_loss"]) - float(g["mse_loss_" + name])) < 1e-5 * max(1.0, float(g["mse_loss_" + name])), name


def test_torch_cpu_oracle_matches_reference(W):
    """The PyTorch-CPU restatement (bench.py's cpu_baseline) against the same reference goldens as the numpy oracle."""
    import torch
    from oracle import torch_cpu as tc
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    g = np.load(os.path.join(GOLDEN, "denoiser_eps_B2_N128_mixed.npz"))
    anchors, variance = _per_point(g)
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ctx = [t_(c) for c in _ctx(g)]
    for t in g["ts"]:
        eps = tc.transformer_net_forward(Wt, t_(g["x"]), torch.full((2,), int(t)), ctx, t_(anchors).transpose(1, 2), t_(variance).transpose(1, 2),
                                         t_(g["valid"]), t_(g["seg"]))
        assert np.abs(eps.numpy() - g[f"eps_t{int(t)}"]).max() < TOL_EPS
    c = np.load(os.path.join(GOLDEN, "chain_T10_B2_N128_mixed.npz"))
    anchors, variance = _per_point(c)
    x = t_(c["traj"][0])
    tb = df.Tables(10)
    for i, t in enumerate(range(9, -1, -1)):
        x, _ = tc.p_sample(tb, Wt, x, t, t_(anchors), [t_(v) for v in _ctx(c)], t_(variance), t_(c["seg"]), t_(c["valid"]), t_(c["step_noise"][i]))
        assert np.abs(x.numpy() - c["traj"][i + 1]).max() < TOL_CHAIN


def test_training_gradient_oracle_matches_reference_autograd():
    """oracle/train.py (torch-CPU autograd over the restated forward) against loss.backward() through the reference's own
    TransformerNet in train mode with dropout 0 (tests/golden/train_grads_*.npz): loss, eps, 77 parameter gradients, ctx grads."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _train_case import check_against_golden, load_case
    from oracle import train
    g, c = load_case("B3_N64_T10")
    r = train.loss_and_grads(**c)
    assert abs(r["loss"] - float(g["loss"])) < 2e-6
    assert np.abs(r["eps"] - g["eps"]).max() < 5e-6
    n, worst = check_against_golden(g, r["grads"], rtol=1e-4, atol=1e-8)
    assert n == 77
    for k in ("d_ctx_code", "d_ctx_mv"):
        assert np.abs(r[k] - g[k]).max() <= 1e-4 * np.abs(g[k]).max() + 1e-9


def test_pointnet_v2_train_oracle_matches_reference_autograd():
    """oracle/pointnet_v2_train.py (F.batch_norm(training=True) restatement) against the reference class in train() mode:
    outputs, running statistics and parameter gradients (tests/golden/pointnet_v2_train_*.npz)."""
    from oracle import pointnet_v2_train as pt
    g = dict(np.load(os.path.join(GOLDEN, "pointnet_v2_train_B5_N160.npz")))
    W = synth.make_pointnet_v2_weights(int(g["weight_seed"]))
    r = pt.outputs_and_grads(W, g["x"], g["attn"], g["dm"], g["dv"])
    assert np.abs(r["m"] - g["m"]).max() < 1e-6 and np.abs(r["v"] - g["v"]).max() < 1e-6
    n = 0
    for key in g:
        if key.startswith("r/"):
            assert np.abs(r["running"][key[2:]] - g[key]).max() < 1e-6
        elif key.startswith("g/"):
            assert np.abs(r["grads"][key[2:]].ravel() - g[key]).max() <= 1e-5 * max(np.abs(g[key]).max(), 1e-3)
            n += 1
        elif key.startswith("gs/"):
            name = key[3:]
            assert np.abs(r["grads"][name].ravel()[g["gi/" + name]] - g[key]).max() <= 1e-5 * np.abs(g[key]).max()
            n += 1
    assert n == 36


def test_prior_loss_oracle_matches_reference_autograd():
    """oracle/prior_loss.py against PartEncoder.get_prior_loss + autograd of the reference encoder (tests/golden/prior_loss_*.npz):
    loss value (with the reference's normalisation constant), d part_code, d logvar, 336 flow-parameter gradients."""
    from oracle import prior_loss as pl
    g = dict(np.load(os.path.join(GOLDEN, "prior_loss_B6.npz")))
    W = synth.make_latent_weights(int(g["weight_seed"]))
    r = pl.loss_and_grads(W, g["part_code"], g["logvar"], g["valid"], prior_var=float(g["prior_var"]), kl_weight=float(g["kl_weight"]))
    assert abs(r["loss"] - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    assert np.abs(r["d_part_code"] - g["d_part_code"]).max() <= 1e-5 * np.abs(g["d_part_code"]).max()
    assert np.abs(r["d_logvar"] - g["d_logvar"]).max() <= 1e-6 * np.abs(g["d_logvar"]).max()
    n = 0
    for key in g:
        if key.startswith("g/"):
            assert np.abs(r["grads"][key[2:]].ravel() - g[key]).max() <= 1e-4 * max(np.abs(g[key]).max(), 1e-30)
            n += 1
        elif key.startswith("gs/"):
            name = key[3:]
            assert np.abs(r["grads"][name].ravel()[g["gi/" + name]] - g[key]).max() <= 1e-4 * max(np.abs(g[key]).max(), 1e-30)
            n += 1
    assert n == 336


def test_golden_manifest():
    """Every committed fixture is listed in tests/golden/MANIFEST.sha256 with the hash of its array contents, nothing is missing or
    extra; `python tests/golden/make_golden.py --all` regenerates all of them and rewrites the manifest."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("golden_manifest", os.path.join(os.path.dirname(__file__), "golden", "manifest.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    missing, extra, changed = m.diff()
    assert not missing and not extra and not changed, (missing, extra, changed)
    assert len(m.read()) >= 34


# ---------------------------------------------------------------------------------- top-level compositions (oracle/forward.py)
def _forward_weights():
    W_enc = dict(synth.make_latent_weights(0))
    W_enc.update({"encoder." + k: v for k, v in synth.make_pointnet_v2_weights(0).items()})
    return W_enc, synth.make_denoiser_weights(0)


def _cmp_dict(pred, expect, tol):
    assert set(map(str, pred)) == set(expect), (sorted(map(str, pred)), sorted(expect))
    worst = 0.0
    for k, v in pred.items():
        ref, got = expect[str(k)], np.asarray(v)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind in "iub":
            assert np.array_equal(got, ref), k
        else:
            err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
            assert err <= tol, (k, err)
            worst = max(worst, err)
    return worst


@pytest.mark.parametrize("tag", ["gen_B2_K2_T10", "sample_B2_K2_T10"])
def test_oracle_anchor_forward_matches_reference_golden(tag):
    """oracle/forward.anchor_forward vs AnchorDiffAE.forward of the reference (every key of the dict Runner.val saves)."""
    from _replay import load_forward_fixture
    from oracle import forward as ofw
    batch, draws, expect, meta = load_forward_fixture(os.path.join(GOLDEN, f"forward_{tag}.npz"))
    W_enc, W_den = _forward_weights()
    nb = {k: v.numpy() for k, v in batch.items()}
    pred, name = ofw.anchor_forward(W_enc, W_den, nb, draws, int(meta["T"]), nb["ref"].shape[1], int(meta["K"]), gen=tag.startswith("gen"),
                                    ret_interval=int(meta["ret_interval"]))
    assert name == str(meta["name"])
    worst = _cmp_dict(pred, expect, 2e-5)
    print(f"oracle forward[{tag}]: worst rel err {worst:.2e}")


def test_oracle_encoder_forward_matches_reference_golden():
    from _replay import load_forward_fixture
    from oracle import forward as ofw
    batch, draws, expect, meta = load_forward_fixture(os.path.join(GOLDEN, "encoder_fwd_B3_N96.npz"))
    W_enc, _ = _forward_weights()
    nb = {k: v.numpy() for k, v in batch.items()}
    draws = list(draws)
    kw = dict(kl_weight=float(meta["kl_weight"]), epoch=int(meta["epoch"]))
    e1 = ofw.encoder_forward(W_enc, nb, draws, **kw)
    noise, idx = ofw.sample_noise(W_enc, nb, draws, int(meta["num"]))
    e2 = ofw.encoder_forward(W_enc, nb, draws, noise=noise, **kw)
    assert not draws
    got = {"ctx0": e1["ctx"][0], "ctx1": e1["ctx"][1], "mean_pp": e1["mean_pp"], "logvar_pp": e1["logvar_pp"], "flag_pp": e1["flag_pp"],
           "part_code": e1["part_code"], "mean": e1["mean"], "logvar": e1["logvar"], "noise": e1["noise"], "sn_noise": noise, "sn_id": idx,
           "k_ctx0": e2["ctx"][0], "k_ctx1": e2["ctx"][1], "k_mean_pp": e2["mean_pp"], "k_logvar_pp": e2["logvar_pp"],
           "k_flag_pp": e2["flag_pp"], "k_fit_loss": e2["losses"]["fit_loss"]}
    got.update({"loss/" + k: np.asarray(v).reshape(expect["loss/" + k].shape) for k, v in e1["losses"].items()})
    worst = _cmp_dict(got, expect, 2e-5)
    print(f"oracle encoder forward: worst rel err {worst:.2e}")

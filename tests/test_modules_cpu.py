"""CPU: the drop-in modules keep the reference's constructor arguments and state_dict layout; when the reference
tree is present (dev container) compare key-for-key and check registry installation."""
import os
import sys

import pytest
import torch

from difffacto_amd import synth
from difffacto_amd.modules import TransformerNet, AnchoredDiffusion

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NET_CFG = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=0.2,
               context_dim=256 + 6, n_class=4, class_cond=True, use_linear=True, cat_params_to_x=True,
               use_checkpoint=False, single_attn=True, cat_class_to_x=True)          # configs/gen_chair.py:54-70
DIFF_CFG = dict(net=NET_CFG, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False,
                rescale_timesteps=False, model_mean_type="epsilon", learn_variance=True, loss_type='mse',
                include_anchors=False, classifier_weight=1., guidance=False, ddim_sampling=False, ddim_nsteps=25,
                ddim_discretize='quad', ddim_eta=1.)                                  # configs/gen_chair.py:52-86


def test_state_dict_matches_synth_layout():
    d = AnchoredDiffusion(num_timesteps=100, **DIFF_CFG)
    sd = d.model.state_dict()
    shapes = dict(synth.denoiser_param_shapes(5))
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k
    W = synth.make_denoiser_weights(0)
    d.model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    assert list(d.state_dict())[0].startswith("model.")


def test_unsupported_options_fail_loudly():
    bad = dict(NET_CFG)
    bad.pop("type")
    bad["single_attn"] = False
    with pytest.raises(NotImplementedError):
        TransformerNet(**bad)
    with pytest.raises(NotImplementedError):
        AnchoredDiffusion(num_timesteps=10, **{**DIFF_CFG, "model_mean_type": "start_x"})
    d = AnchoredDiffusion(num_timesteps=1000, **{**DIFF_CFG, "ddim_sampling": True, "ddim_nsteps": 25, "ddim_discretize": "quad"})
    assert d.steps[:4] == [0, 1, 5, 12] and len(d.steps) == 25 and d.steps[-1] == 800   # anchored_diffusion.py:120-122


def test_cpu_forward_is_rejected():
    args = dict(NET_CFG)
    args.pop("type")
    net = TransformerNet(**args)
    with pytest.raises(RuntimeError, match="CPU not supported|HIP device"):
        net(torch.zeros(1, 3, 32), torch.zeros(1, dtype=torch.long), [torch.zeros(1, 256, 4), torch.ones(1, 6, 4)],
            valid_id=torch.ones(1, 4), anchor_assignment=torch.zeros(1, 32, dtype=torch.int32))


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/difffacto"), reason="reference tree not present")
def test_keys_and_registry_against_reference():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_import
    model, cfg = ref_import.build_reference_model(num_timesteps=10)
    ref_sd = model.diffusion.state_dict()
    mine = AnchoredDiffusion(num_timesteps=10, **DIFF_CFG).state_dict()
    assert set(ref_sd) == set(mine)
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(mine[k].shape), k
    # registry installation: the reference's build_from_cfg now yields the libdfx-backed classes
    import difffacto_amd
    from difffacto.utils.registry import NETS, DIFFUSIONS, build_from_cfg
    saved = (NETS._modules["TransformerNet"], DIFFUSIONS._modules["AnchoredDiffusion"], dict(sys.modules))
    try:
        assert difffacto_amd.install() is True
        built = build_from_cfg(dict(cfg.model["diffusion"], num_timesteps=10), DIFFUSIONS)
        assert isinstance(built, AnchoredDiffusion) and isinstance(built.model, TransformerNet)
        import pointnet2_ops
        assert pointnet2_ops.pointnet2_utils.gather_operation.__self__.__module__.startswith("difffacto_amd")
    finally:
        NETS._modules["TransformerNet"], DIFFUSIONS._modules["AnchoredDiffusion"] = saved[0], saved[1]
        for k in ("pointnet2_ops", "pointnet2_ops.pointnet2_utils", "pointnet2_ops.pointnet2_modules"):
            sys.modules[k] = saved[2][k]


ENC_CFG = dict(
    encoder=dict(type='PointNetV2', zdim=256, point_dim=3, per_part_mlp=True),
    part_aligner=dict(type="PartAlignerTransformer", in_channels=256, out_channels=6, n_class=4, d_head=32, depth=5,
                      n_heads=8, dropout=0., use_checkpoint=False, use_linear=True, class_cond=True, single_attn=True,
                      add_class_cond=True, cimle=True, noise_scale=100, cond_noise_type=0),
    n_class=4, kl_weight=0, fit_loss_type=4, fit_loss_weight=1.0, use_flow=True, latent_flow_depth=14,
    latent_flow_hidden_dim=256, include_z=False, include_part_code=True, include_params=True, use_gt_params=False,
    kl_weight_annealing=False, gen=True, prior_var=1.0)   # configs/gen_chair.py:6-47


TOP_CFG = dict(sampler=dict(type='Uniform'), num_anchors=4, num_timesteps=100, npoints=2048, gen=True, cimle=True, cimle_sample_num=1, ret_traj=True,
               ret_interval=10, forward_sample=False, drift_anchors=False, interpolate=False, save_weights=False)   # configs/gen_chair.py:88-104


def model_cfg(**overrides):
    """`cfg.model` of the unmodified configs/gen_chair.py without its 'type' key (test_install_full_... checks that against the reference's own
    config loader), with the knobs the golden fixtures were generated under (num_timesteps, npoints, cimle_sample_num, ret_interval, ...) replaced:
    what the -m gpu composition tests construct networks.AnchorDiffAE from."""
    net = overrides.pop("net", {})
    diffusion = dict(type="AnchoredDiffusion", **{**DIFF_CFG, "net": dict(DIFF_CFG["net"], **net)})
    return {**TOP_CFG, "encoder": dict(type="PartEncoderForTransformerDecoder", **ENC_CFG), "diffusion": diffusion, **overrides}


def _plain(c):
    return {k: _plain(v) for k, v in c.items()} if hasattr(c, "items") else c


def test_encoder_mirror_state_dict_keys_and_unsupported_options():
    from difffacto_amd.encoders import PartEncoderForTransformerDecoder, PartAlignerTransformer
    enc = PartEncoderForTransformerDecoder(**ENC_CFG)
    W = synth.make_latent_weights(0)
    W.update({"encoder." + k: v for k, v in synth.make_pointnet_v2_weights(0).items()})
    sd = {k: v for k, v in enc.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert set(sd) == set(W)
    for k, v in sd.items():
        assert tuple(v.shape) == W[k].shape, k
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    with pytest.raises(NotImplementedError):
        PartAlignerTransformer(256, 8, 32, 6, depth=5, use_linear=True, single_attn=True, add_class_cond=True, cimle=True,
                               cond_noise_type=2)
    with pytest.raises(NotImplementedError):
        PartEncoderForTransformerDecoder(**{**ENC_CFG, "selective_noise_sampling": True})


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/difffacto"), reason="dev container only")
def test_encoder_mirror_matches_reference_state_dict():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import ref_import
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model, _ = ref_import.build_reference_model("gen_chair.py", 10)
    from difffacto_amd.encoders import PartEncoderForTransformerDecoder
    enc = PartEncoderForTransformerDecoder(**ENC_CFG)
    ref = {k: tuple(v.shape) for k, v in model.encoder.state_dict().items()}
    assert ref == {k: tuple(v.shape) for k, v in enc.state_dict().items()}       # incl. encoder.* (PointNetV2): strict loading
    enc.load_state_dict(model.encoder.state_dict(), strict=True)
    from difffacto_amd.encoders import attach
    mirror = attach(model.encoder)           # construction only (the first sample_latents call needs the GPU)
    assert {k for k in mirror.state_dict() if not k.startswith("encoder.")} == {k for k in ref if not k.startswith("encoder.")}
    assert mirror.part_aligner.noise_scale == 100
    assert model.encoder.sample_latents.__name__ == "sample_latents"


def test_point_padding_helpers_of_the_engine():
    """Host logic behind 'any N like the reference': the kernels take N % 32 == 0, the engine pads with zeros / the shape's first
    label and cuts the padding off (points are independent).  The helpers are plain tensor code: checked here without a GPU."""
    import torch
    from difffacto_amd.engine import DenoiserEngine as E
    assert [E._pad(n) for n in (0, 1, 31, 32, 33, 100, 2048, 2080)] == [0, 31, 1, 0, 31, 28, 0, 0]
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    xp = E._pad_last(x, 3)
    assert xp.shape == (2, 3, 8) and torch.equal(xp[..., :5], x) and float(xp[..., 5:].abs().sum()) == 0.0
    assert E._pad_last(None, 3) is None
    seg = torch.tensor([[2, 1, 0, 3, 3], [1, 1, 1, 0, 2]], dtype=torch.int32)
    sp = E._pad_seg(seg, 3)
    assert sp.shape == (2, 8) and sp.is_contiguous() and torch.equal(sp[:, :5], seg)
    assert sp[0, 5:].tolist() == [2, 2, 2] and sp[1, 5:].tolist() == [1, 1, 1]      # a label that is present in that shape


def test_pointnet_v2_rejects_cpu_tensors_in_every_mode():
    """No torch-layer fallback behind the native part encoder: CPU tensors raise the reference op's
    'CPU not supported' in eval / no_grad, eval / grad and train() alike."""
    import torch
    from difffacto_amd.encoders import PointNetV2
    enc = PointNetV2(zdim=256, num_anchors=4, per_part_mlp=True)
    x, attn = torch.zeros(2, 64, 3), torch.ones(2, 64, 4)
    for train in (False, True):
        enc.train(train)
        for grad in (False, True):
            with torch.set_grad_enabled(grad), pytest.raises(RuntimeError, match="CPU not supported"):
                enc(x, attn)


def test_parameter_generation_is_per_parameter_set():
    """an optimiser step on one parameter set must not invalidate the packed-weight caches of another (they key on
    training.generation_of(their own parameters), which Adam.step bumps per stepped tensor)."""
    import torch
    from difffacto_amd import training
    a, b = [torch.zeros(3), torch.zeros(2, 2)], [torch.zeros(5)]
    ga, gb, ga0 = training.generation_of(a), training.generation_of(b), training.generation_of(a[:1])
    training._bump_generation(a)
    g1 = training.generation_of(a)
    training._bump_generation(a)
    # every step of a's optimiser moves a's generation (and that of each of its tensors), never b's
    assert training.generation_of(a) > g1 > ga and training.generation_of(b) == gb and training.generation_of(a[:1]) > ga0


def test_generation_counter_follows_storage_aliases():
    """packed-weight caches key on training.generation_of(their parameters); an update through an ALIAS of a parameter
    (second Parameter / view on the same storage) must move it too, an update of unrelated tensors must not."""
    import torch
    from difffacto_amd import training
    p, q = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))
    alias = torch.nn.Parameter(p.data)          # same storage, different tensor object
    view = p.data[2:6]
    g0p, g0q = training.generation_of([p]), training.generation_of([q])
    training._bump_generation([alias])
    assert training.generation_of([p]) > g0p and training.generation_of([q]) == g0q
    g1p = training.generation_of([p])
    training._bump_generation([view])
    assert training.generation_of([p]) > g1p and training.generation_of([q]) == g0q
    training._bump_generation([q])
    assert training.generation_of([q]) > g0q


def test_bench_evidence_helpers_read_the_newest_committed_profiles():
    """bench.py's roofline.traffic names the profile file and round it came from (newest committed rNN_traffic.json),
    and the power-limited ceiling is parsed from the newest committed micro-benchmark run instead of a hard-coded constant."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    traffic, src = b.measured_traffic(1000, 128, 2048)
    # round 5: fixed + per_step * T from PMC passes at two chain lengths (profiles/r05_write_size_scaling.txt: the writes do not grow with T at all;
    # rounds 1-4 multiplied ONE T = 20 launch by T / 20 and reported 4.2 GB) -> ~0.19 GB per T = 1000 launch
    assert traffic is not None and 1e8 < traffic < 1e9 and src["file"].startswith("profiles/r") and src["file"].endswith("_traffic.json")
    assert int(src["round"][1:]) >= 5 and "not measured in this run" in src["note"] and src["bytes_per_launch_fixed"] > 5e7
    assert abs(b.measured_traffic(40, 128, 2048)[0] - (src["bytes_per_launch_fixed"] + 40 * src["bytes_per_step"])) < 1.0
    assert b.measured_traffic(1000, 7, 2048) == (None, None)          # no profile for that batch: no number
    c = b.power_limited_ceiling()
    assert c["source"].startswith("profiles/r") and 1500 < c["bare_mfma_tflops"] < 2500 and 1000 < c["kernel_mix_tflops"] < c["refill_per_mfma_tflops"] < c["bare_mfma_tflops"]
    assert abs(b.flops_per_step(2048) - 4.733899e9) < 1e4


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/difffacto"), reason="dev container only")
def test_reference_forward_through_install_and_attach_dry_run(monkeypatch):
    """The drop-in protocol end to end WITHOUT a GPU: the reference's OWN AnchorDiffAE (built by its registry after
    difffacto_amd.install(), encoder accelerated with encoders.attach()) runs its gen branch (anchor_gen.py:1034-1084); the two libdfx
    handles are replaced by numpy-oracle stand-ins (tests/_oracle_backend.py, test-only).  Checks:
      * the kwargs the reference passes (decode -> p_sample_loop_progressive(shape, anchors=, variance=, ctx=, noise=None,
        anchor_assignment=, valid_id=, device=, progress=) -> p_sample) are accepted, and the dict it builds equals the golden the
        unmodified reference produced (forward_gen_B2_K2_T10.npz) when the recorded draws are replayed at the same sites;
      * seed-less sampling is FRESH per decode() call — ONE key per loop, a new one per call — and replays under torch.manual_seed."""
    import contextlib
    import io
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import ref_import
    from _oracle_backend import OracleEngine, OracleLatentSampler
    from _replay import load_forward_fixture, replay_draws
    import difffacto_amd
    from difffacto_amd import encoders, modules
    ref_import.import_reference()
    from difffacto.config.config import init_cfg, get_cfg
    from difffacto.utils.registry import NETS, DIFFUSIONS, MODELS, build_from_cfg

    batch, draws, expect, meta = load_forward_fixture(os.path.join(ROOT, "tests", "golden", "forward_gen_B2_K2_T10.npz"))
    T, K, N = int(meta["T"]), int(meta["K"]), batch["ref"].shape[1]
    saved = (NETS._modules["TransformerNet"], DIFFUSIONS._modules["AnchoredDiffusion"], dict(sys.modules))
    try:
        assert difffacto_amd.install() is True
        init_cfg(os.path.join(ref_import.REF_ROOT, "configs", "gen_chair.py"))
        cfg = get_cfg()
        cfg.model["num_timesteps"] = T
        with contextlib.redirect_stdout(io.StringIO()):
            model = build_from_cfg(cfg.model, MODELS).eval()
        assert type(model).__module__.startswith("difffacto.") and isinstance(model.diffusion, modules.AnchoredDiffusion)
        model.npoints, model.cimle_sample_num, model.ret_traj, model.ret_interval = N, K, True, int(meta["ret_interval"])
        model.diffusion.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()})
        sd = model.encoder.state_dict()
        for k, v in synth.make_latent_weights(0).items():
            sd[k] = torch.from_numpy(v.copy())
        for k, v in synth.make_pointnet_v2_weights(0).items():
            sd["encoder." + k] = torch.from_numpy(v.copy())
        model.encoder.load_state_dict(sd)
        encoders.attach(model.encoder)

        engines = {}

        def fake_engine(net):
            key = id(net)
            if key not in engines:
                engines[key] = OracleEngine({k: v.detach() for k, v in net.named_parameters()}, net._dfx_T, *net._dfx_betas)
            return engines[key]

        monkeypatch.setattr(modules.TransformerNet, "engine", fake_engine)
        monkeypatch.setattr(encoders, "LatentSampler", OracleLatentSampler)
        monkeypatch.setattr(torch.Tensor, "cuda", lambda t, *a, **k: t, raising=False)    # hard-coded .cuda() in the reference's prior loss

        chain_at = 3
        OracleEngine.replay_steps = list(draws[chain_at + 1:chain_at + T + 1])
        with replay_draws(draws[:chain_at + 1] + draws[chain_at + T + 1:]) as queue, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            out = model(batch, device="cpu", epoch=0)
        assert not queue and not OracleEngine.replay_steps
        OracleEngine.replay_steps = None
        pred, name = out[0]
        assert name == str(meta["name"]) and set(map(str, pred)) == set(expect)
        for k, v in pred.items():
            ref, got = expect[str(k)], v.numpy()
            assert got.shape == ref.shape, k
            if ref.dtype.kind in "iub":
                assert np.array_equal(got, ref), k
            else:
                assert float(np.abs(got - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), k

        # ---- fresh noise per decode() call of the REFERENCE (it passes no seed: anchor_gen.py:149-158) ----
        pc, mean, logvar, valid = synth.make_latents(2, seed=5)
        t_ = torch.from_numpy
        ctx = [t_(pc), torch.cat([t_(mean), torch.exp(t_(logvar))], 1)]
        seg = t_(synth.make_seg_mask(valid, 32))
        idx = seg.long()[:, None, :].expand(-1, 3, -1)
        anchors, variance = torch.gather(ctx[1][:, :3], 2, idx), torch.gather(ctx[1][:, 3:], 2, idx)
        kw = dict(ctx=ctx, variance=variance, anchor_assignments=seg.to(torch.int32), valid_id=t_(valid), device="cpu")
        OracleEngine.seeds_seen = []
        torch.manual_seed(11)
        with torch.no_grad():
            a, b = model.decode(anchors, **kw)["pred"], model.decode(anchors, **kw)["pred"]
        seeds = list(OracleEngine.seeds_seen)
        assert len(seeds) == 2 * T and len(set(seeds[:T])) == 1 and len(set(seeds[T:])) == 1 and seeds[0] != seeds[T]   # one key per loop
        assert not torch.equal(a, b)
        torch.manual_seed(11)
        with torch.no_grad():
            a2 = model.decode(anchors, **kw)["pred"]
        assert torch.equal(a, a2)
    finally:
        OracleEngine.replay_steps = None
        NETS._modules["TransformerNet"], DIFFUSIONS._modules["AnchoredDiffusion"] = saved[0], saved[1]
        for k in ("pointnet2_ops", "pointnet2_ops.pointnet2_utils", "pointnet2_ops.pointnet2_modules"):
            sys.modules[k] = saved[2][k]


@pytest.mark.parametrize("tag,chain_at", [("gen_B2_K2_T10", 3), ("sample_B2_K2_T10", 4)])
def test_anchor_diff_ae_mirror_bookkeeping_on_oracle_backends(tag, chain_at, monkeypatch):
    """networks.AnchorDiffAE.forward (the mirror's dict bookkeeping, draw order, K-fold regrouping) with every libdfx handle replaced by
    the numpy oracle (test-only stand-ins): equals the golden of the reference's AnchorDiffAE.forward.  The GPU twin of this test
    (tests/test_gpu_forward.py) runs the same comparison through the real kernels."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from _oracle_backend import OracleEngine, OracleLatentSampler
    from _replay import load_forward_fixture, replay_draws
    from difffacto_amd import encoders, modules, training
    from difffacto_amd.networks import AnchorDiffAE
    from oracle import pointnet_v2 as opv
    from oracle import prior_loss as opl
    batch, draws, expect, meta = load_forward_fixture(os.path.join(ROOT, "tests", "golden", f"forward_{tag}.npz"))
    T, K, N = int(meta["T"]), int(meta["K"]), batch["ref"].shape[1]
    m = AnchorDiffAE(encoder=dict(type="PartEncoderForTransformerDecoder", **ENC_CFG), diffusion=dict(type="AnchoredDiffusion", **DIFF_CFG),
                     sampler=dict(type="Uniform"), num_anchors=4, num_timesteps=T, npoints=N, gen=tag.startswith("gen"), cimle=True,
                     cimle_sample_num=K, ret_traj=True, ret_interval=int(meta["ret_interval"]))
    W = {"diffusion.model." + k: v for k, v in synth.make_denoiser_weights(0).items()}
    W.update({"encoder." + k: v for k, v in synth.make_latent_weights(0).items()})
    W.update({"encoder.encoder." + k: v for k, v in synth.make_pointnet_v2_weights(0).items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    m.eval()
    engines = {}
    monkeypatch.setattr(modules.TransformerNet, "engine", lambda net: engines.setdefault(id(net), OracleEngine(
        {k: v.detach() for k, v in net.named_parameters()}, net._dfx_T, *net._dfx_betas)))
    monkeypatch.setattr(encoders, "LatentSampler", OracleLatentSampler)
    Wpn = synth.make_pointnet_v2_weights(0)
    monkeypatch.setattr(encoders.PointNetV2, "forward", lambda self, x, a: tuple(map(torch.from_numpy, opv.forward(Wpn, x.numpy(), a.numpy()))))

    def prior(params, part_code, logvar, valid, depth=14, hidden=256, prior_var=1.0, kl_weight=5e-4, stream=None):
        return opl.prior_loss({k: v.detach() for k, v in params.items()}, part_code, logvar, valid, depth=depth, prior_var=prior_var, kl_weight=kl_weight)
    monkeypatch.setattr(training, "prior_loss", prior)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: __import__("contextlib").nullcontext())
    chain = draws[chain_at:chain_at + T + 1]
    orig = m.decode
    m.decode = lambda *a, **k: orig(*a, x_T_noise=torch.from_numpy(chain[0]), step_noise=torch.from_numpy(np.stack(chain[1:])), **k)
    with replay_draws(draws[:chain_at] + draws[chain_at + T + 1:]) as queue:
        out = m(batch, device="cpu", epoch=0)
    assert not queue
    pred, name = out[0]
    assert name == str(meta["name"]) and set(map(str, pred)) == set(expect), (sorted(map(str, pred)), sorted(expect))
    for k, v in pred.items():
        ref, got = expect[str(k)], v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if ref.dtype.kind in "iub":
            assert np.array_equal(got, ref), k
        else:
            assert float(np.abs(got - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max())), k


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/difffacto"), reason="dev container only")
@pytest.mark.parametrize("cfg_name", ["gen_chair.py", "gen_airplane.py", "gen_car.py", "gen_lamp.py", "train_chair_stage1.py", "train_chair_stage2.py"])
def test_install_full_builds_the_mirror_from_the_unmodified_configs(cfg_name):
    """INTEGRATION.md §2's "take the whole model" binding: after `install(full=True)` the REFERENCE's own `init_cfg` +
    `build_from_cfg(cfg.model, MODELS)` (utils/registry.py:24-46) on an unmodified shipped config yields `networks.AnchorDiffAE` whose `state_dict`
    has the reference model's keys and shapes, and a reference-made checkpoint loads `strict=True` — also through the shape-checking,
    `strict=False` protocol of `Runner.load` (runner.py:505-520), with nothing dropped and nothing left unloaded."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import contextlib
    import io
    import ref_import
    import difffacto_amd
    from difffacto_amd import networks, encoders, modules
    with contextlib.redirect_stdout(io.StringIO()):
        ref_model, cfg = ref_import.build_reference_model(cfg_name)
    assert type(ref_model).__module__.startswith("difffacto.")
    ref_sd = ref_model.state_dict()
    from difffacto.utils.registry import NETS, DIFFUSIONS, MODELS, ENCODERS, SAMPLERS, build_from_cfg
    regs = (NETS, DIFFUSIONS, MODELS, ENCODERS, SAMPLERS)
    saved = [dict(r._modules) for r in regs]
    saved_mods = {k: sys.modules.get(k) for k in ("pointnet2_ops", "pointnet2_ops.pointnet2_utils", "pointnet2_ops.pointnet2_modules")}
    try:
        assert difffacto_amd.install(full=True) is True
        with contextlib.redirect_stdout(io.StringIO()):
            mine = build_from_cfg(cfg.model, MODELS)
        assert type(mine) is networks.AnchorDiffAE
        if cfg_name == "gen_chair.py":       # the dict the -m gpu composition tests build the mirror from IS this config's cfg.model
            assert _plain(cfg.model) == {"type": "AnchorDiffAE", **model_cfg()}
        if cfg_name == "train_chair_stage2.py":   # stage 2 = gen_chair's model up to these keys (the stage2_step golden was made on gen_chair's)
            got, base = _plain(cfg.model), {"type": "AnchorDiffAE", **model_cfg()}
            assert got.pop("save_dir") and got.pop("ret_interval") == 1 and "cimle_sample_num" not in got
            assert got["encoder"]["part_aligner"].pop("noise_scale") == 50
            base.pop("ret_interval"), base.pop("cimle_sample_num"), base["encoder"]["part_aligner"].pop("noise_scale")
            assert got == base
        assert type(mine.encoder) is encoders.PartEncoderForTransformerDecoder and type(mine.encoder.encoder) is encoders.PointNetV2
        assert ENCODERS.get("PointNet2SSG") is encoders.PointNet2SSG and ENCODERS.get("PointNet2MSG") is encoders.PointNet2MSG   # (registered, unused by the configs)
        assert type(mine.diffusion) is modules.AnchoredDiffusion and type(mine.diffusion.model) is modules.TransformerNet
        my_sd = mine.state_dict()
        assert list(my_sd) == list(ref_sd) or set(my_sd) == set(ref_sd)
        assert len(ref_sd) == (473 if cfg_name == "train_chair_stage1.py" else 547)   # stage 1 has no part aligner (train_chair_stage1.py)
        for k, v in ref_sd.items():
            assert tuple(my_sd[k].shape) == tuple(v.shape) and my_sd[k].dtype == v.dtype, k
        assert sum(p.numel() for p in mine.parameters()) == sum(p.numel() for p in ref_model.parameters())
        mine.load_state_dict(ref_sd, strict=True)
        for k, v in mine.state_dict().items():
            assert torch.equal(v, ref_sd[k]), k
        # Runner.load's protocol on a checkpoint written by DataParallel-wrapped training ("module." prefix stripped by the caller, runner.py:503)
        with contextlib.redirect_stdout(io.StringIO()):
            again = build_from_cfg(cfg.model, MODELS)
        ckpt = {"module." + k: v.clone() for k, v in ref_sd.items()}
        state = {k.replace("module.", ""): v for k, v in ckpt.items()}
        target = again.state_dict()
        dropped = [k for k in state if k not in target]
        reshaped = [k for k in state if k in target and state[k].shape != target[k].shape]
        missing = [k for k in target if k not in state]
        assert not dropped and not reshaped and not missing
        res = again.load_state_dict(state, strict=False)
        assert not res.missing_keys and not res.unexpected_keys
        assert all(torch.equal(v, ref_sd[k]) for k, v in again.state_dict().items())
    finally:
        for r, s in zip(regs, saved):
            r._modules.clear()
            r._modules.update(s)
        for k, v in saved_mods.items():
            if v is not None:
                sys.modules[k] = v

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
        AnchoredDiffusion(num_timesteps=10, **{**DIFF_CFG, "ddim_sampling": True})


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

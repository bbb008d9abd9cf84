"""Dev-container-only helper: import the reference Python model (/root/reference) on CPU.

Never shipped to / used on the GPU box.  Installs the six stub modules described in
SURVEY.md §8(c) so that `import difffacto` works without CUDA extensions, then exposes
`build_reference_model(cfg_path)`.

Used only by tests/golden/make_golden.py (fixture generation) and by the dev-only
cross-check tests that are skipped when /root/reference is absent.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("DFX_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "python", "difffacto"))


def _install_stubs():
    import torch

    if "pointnet2_ops" in sys.modules and getattr(sys.modules["pointnet2_ops"], "_dfx_stub", False):
        return
    p2 = types.ModuleType("pointnet2_ops")
    p2._dfx_stub = True
    p2u = types.ModuleType("pointnet2_ops.pointnet2_utils")

    def gather_operation(features, idx):
        # pure-torch statement of SRC/sampling_gpu.cu:8-20 (out[b,c,j] = points[b,c,idx[b,j]])
        C = features.shape[1]
        return torch.gather(features, 2, idx.long()[:, None].expand(-1, C, -1))

    def furthest_point_sample(xyz, npoint):
        raise RuntimeError("stub: FPS is not on the CPU reference path")

    p2u.gather_operation = gather_operation
    p2u.furthest_point_sample = furthest_point_sample
    p2m = types.ModuleType("pointnet2_ops.pointnet2_modules")

    class _Placeholder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    p2m.PointnetSAModule = _Placeholder
    p2m.PointnetSAModuleMSG = _Placeholder
    p2m.PointnetFPModule = _Placeholder
    p2.pointnet2_utils = p2u
    p2.pointnet2_modules = p2m
    sys.modules["pointnet2_ops"] = p2
    sys.modules["pointnet2_ops.pointnet2_utils"] = p2u
    sys.modules["pointnet2_ops.pointnet2_modules"] = p2m

    tbx = types.ModuleType("tensorboardX")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    tbx.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tbx

    ply = types.ModuleType("plyfile")
    ply.PlyData = object
    ply.PlyElement = object
    sys.modules["plyfile"] = ply

    tde = types.ModuleType("torchdiffeq")
    tde.odeint_adjoint = None
    tde.odeint = None
    sys.modules["torchdiffeq"] = tde

    sys.modules["emd"] = types.ModuleType("emd")
    sys.modules["chamfer"] = types.ModuleType("chamfer")


def import_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present")
    _install_stubs()
    pth = os.path.join(REF_ROOT, "python")
    if pth not in sys.path:
        sys.path.insert(0, pth)
    import difffacto  # noqa: F401

    return difffacto


def build_reference_model(cfg_name="gen_chair.py", num_timesteps=None):
    """Build AnchorDiffAE from a shipped config (CPU, eval)."""
    import_reference()
    from difffacto.config.config import init_cfg, get_cfg
    from difffacto.utils.registry import build_from_cfg, MODELS

    init_cfg(os.path.join(REF_ROOT, "configs", cfg_name))
    cfg = get_cfg()
    if num_timesteps is not None:
        cfg.model["num_timesteps"] = num_timesteps
    model = build_from_cfg(cfg.model, MODELS)
    model.eval()
    return model, cfg

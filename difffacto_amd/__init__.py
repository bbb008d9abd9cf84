"""difffacto_amd — MI355X-native (gfx950) implementation of DiffFacto's reverse-diffusion sampling hot path.

Hand-written HIP kernels behind a C-ABI (`include/dfx.h`, `libdfx.so`) + thin host mirrors of the reference's
`pointnet2_ops` and denoiser / diffusion module API.  See DESIGN.md and INTEGRATION.md.
"""
import sys

__version__ = "0.1.0"


def install(register_in_reference=True, full=False):
    """Make this package the provider of the reference's names.

    * ``sys.modules['pointnet2_ops']`` (+ ``.pointnet2_utils`` / ``.pointnet2_modules``) -> ``difffacto_amd.pointnet2_ops``
      so that ``from pointnet2_ops.pointnet2_utils import gather_operation`` (part_encoders.py:9, anchor_gen.py:9,
      utils/misc.py:7, shapenet_seg.py:13) resolves to the HIP kernels;
    * if the reference package ``difffacto`` is importable and ``register_in_reference``: replace
      ``NETS['TransformerNet']`` and ``DIFFUSIONS['AnchoredDiffusion']`` (utils/registry.py:49-63) by the libdfx-backed
      classes, so ``configs/gen_*.py`` build them through ``build_from_cfg`` unchanged.  By default the encoder registry
      entries are left alone (the reference's own encoder then serves every option it has); accelerate the generation entry point
      of a built model with ``difffacto_amd.encoders.attach(model.encoder)``;
    * ``full=True``: also ``MODELS['AnchorDiffAE']``, ``ENCODERS['PartEncoderForTransformerDecoder' / 'PointNetV2' /
      'PartAlignerTransformer' / 'PointNet2SSG' / 'PointNet2MSG']`` and ``SAMPLERS['Uniform']`` -> the mirrors of ``networks.py`` / ``encoders.py`` (the shipped
      gen_* / train_*_stage1 configurations run end to end on libdfx; any other option raises NotImplementedError).
    """
    from . import pointnet2_ops
    sys.modules["pointnet2_ops"] = pointnet2_ops
    sys.modules["pointnet2_ops.pointnet2_utils"] = pointnet2_ops.pointnet2_utils
    sys.modules["pointnet2_ops.pointnet2_modules"] = pointnet2_ops.pointnet2_modules
    if register_in_reference:
        try:
            from difffacto.utils.registry import NETS, DIFFUSIONS  # the reference package, if present
        except Exception:
            return False
        from .modules import TransformerNet, AnchoredDiffusion
        NETS._modules["TransformerNet"] = TransformerNet
        DIFFUSIONS._modules["AnchoredDiffusion"] = AnchoredDiffusion
        if full:
            from difffacto.utils.registry import MODELS, ENCODERS, SAMPLERS
            from . import encoders, networks
            MODELS._modules["AnchorDiffAE"] = networks.AnchorDiffAE
            SAMPLERS._modules["Uniform"] = networks.Uniform
            for name in ("PartEncoderForTransformerDecoder", "PointNetV2", "PartAlignerTransformer", "PointNet2SSG", "PointNet2MSG"):
                ENCODERS._modules[name] = getattr(encoders, name)
        return True
    return False

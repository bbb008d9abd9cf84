"""Drop-in for the reference ``pointnet2_ops`` package (pointnet2_ops_lib/pointnet2_ops), backed by
libdfx's hand-written gfx950 kernels instead of the CUDA ``_ext`` module.

``import difffacto_amd.pointnet2_ops as pointnet2_ops`` or ``difffacto_amd.install()`` (which puts
it in ``sys.modules['pointnet2_ops']``) gives callers the same names:
``pointnet2_ops.pointnet2_utils`` and ``pointnet2_ops.pointnet2_modules``.
"""
from . import pointnet2_utils, pointnet2_modules  # noqa: F401

__version__ = "3.0.0"  # the reference's pointnet2_ops/_version.py value

"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/dfx.h declares; host-side modules keep the reference's names and state_dict layout.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from difffacto_amd import build
    return build.build(verbose=False)


def _declared_symbols(headers=("dfx.h", "dfx_debug.h")):
    """dfx.h = the drop-in ABI; dfx_debug.h = test / tuning hooks (A/B switches, sweeps), kept out of the product header."""
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(dfx_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_product_header_has_no_debug_switches():
    assert not [n for n in _declared_symbols(("dfx.h",)) if n.startswith("dfx_debug")]
    assert all(n.startswith("dfx_debug") for n in _declared_symbols(("dfx_debug.h",)))


def test_header_symbols_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"libdfx.so does not export {n}"


def test_ffi_signatures_cover_header(built_lib):
    from difffacto_amd import _ffi
    assert sorted(_ffi.SIGNATURES) == _declared_symbols()
    L = _ffi.lib()
    assert L.dfx_version() >= 100 and L.dfx_abi_version() == _ffi.DFX_ABI_VERSION
    assert L.dfx_chain_num_snapshots(100, 10) == 10
    assert L.dfx_chain_num_snapshots(10, 3) == 3
    assert L.dfx_fps_max_resident() >= 8192


def test_error_codes_without_gpu(built_lib):
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    from difffacto_amd import _ffi
    L = _ffi.lib()
    rc = L.dfx_gather_points_f32(None, None, None, -1, 1, 1, 1, None)
    assert rc == -1 and b"negative" in L.dfx_last_error()
    assert L.dfx_gather_points_f32(None, None, None, 0, 3, 5, 0, None) == 0          # empty input is a no-op
    assert L.dfx_furthest_point_sampling_f32(None, None, None, 0, 0, 0, None) == 0
    assert L.dfx_denoise_eps(None, None, None, None, 0, None, 1, 32, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from difffacto_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.DfxLibraryError):
        _ffi.lib()


def test_pointnet2_api_surface():
    from difffacto_amd.pointnet2_ops import pointnet2_utils as pu, pointnet2_modules as pm
    for name in ["furthest_point_sample", "gather_operation", "three_nn", "three_interpolate",
                 "grouping_operation", "ball_query", "QueryAndGroup", "GroupAll"]:
        assert hasattr(pu, name)
    sa = pm.PointnetSAModule(npoint=512, radius=0.2, nsample=64, mlp=[4, 64, 64, 128], use_xyz=True)
    keys = list(sa.state_dict().keys())
    assert keys[0] == "mlps.0.0.weight" and sa.state_dict()["mlps.0.0.weight"].shape == (64, 7, 1, 1)
    msg = pm.PointnetSAModuleMSG(npoint=128, radii=[0.1, 0.2], nsamples=[16, 32], mlps=[[3, 8], [3, 16]])
    assert len(msg.groupers) == 2
    fp = pm.PointnetFPModule(mlp=[16, 8])
    assert "mlp.0.weight" in fp.state_dict()


def test_cpu_tensors_rejected_like_reference():
    import torch
    from difffacto_amd.pointnet2_ops import pointnet2_utils as pu
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pu.gather_operation(torch.zeros(1, 3, 4), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pu.furthest_point_sample(torch.zeros(1, 8, 3), 4)


def test_tile_major_row_addressing_is_a_permutation_and_layout_independent(built_lib):
    """train_ff_fused.h's RowMap (host-compiled from the kernels' own code): inside a 32-point tile the B-operand-layout accessors and the
    accumulator-layout accessors must address the SAME float for every (point, channel); the tile-major layout must be a permutation of the
    tile's 4096 floats made of whole 16-byte chunks, and the row-major one the identity."""
    import numpy as np
    lib = ctypes.CDLL(built_lib)
    lib.dfx_debug_rowmap.restype = None
    lib.dfx_debug_rowmap.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for tiled in (0, 1):
        b = np.full((32, 128), -1, dtype=np.int32)
        a = np.full((32, 128), -1, dtype=np.int32)
        lib.dfx_debug_rowmap(tiled, b.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(a, b)
        assert sorted(b.ravel().tolist()) == list(range(4096))
        if not tiled:
            assert np.array_equal(b, np.arange(4096, dtype=np.int32).reshape(32, 128))
        else:
            # a wavefront's load instruction = one (c, u, half) block: 64 lanes x 16 bytes contiguous (whole cache lines)
            chunks = b.reshape(32, 32, 4)
            assert np.all(chunks[:, :, 1:] - chunks[:, :, :1] == np.arange(1, 4)) and np.all(chunks[:, :, 0] % 4 == 0)
            blocks = (b // 256).reshape(32, 128)
            for blk in range(16):
                pts, chs = np.nonzero(blocks == blk)
                assert len(pts) == 256 and len(set(pts.tolist())) == 32   # every block holds 8 channels of all 32 points


def test_epilogue_batch_statistics_merge_is_as_good_as_two_passes(built_lib):
    """The (count, mean, M2) arithmetic of k_lin_wide_lds<.., LM_STATS> / k_bn_merge, host-compiled from the kernels' own stats_merge
    (dfx_debug_stats_merge): pieces of 16 values, merged left to right, against float64 mean / variance — also for a column whose mean is
    300 standard deviations away from zero (where E[x^2] - E[x]^2 in fp32 would keep two digits) — and with a ragged last piece."""
    import numpy as np
    lib = ctypes.CDLL(built_lib)
    lib.dfx_debug_stats_merge.restype = None
    lib.dfx_debug_stats_merge.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.Generator(np.random.PCG64(5))
    for n, chunk, mean, std in ((262144, 16, 0.3, 1.0), (8192, 16, 300.0, 1.0), (9000, 16, -2.0, 0.01), (1000, 7, 5.0, 2.0), (5, 16, 1.0, 1.0)):
        v = (mean + std * rng.standard_normal(n)).astype(np.float32)
        out = np.zeros(3, dtype=np.float32)
        lib.dfx_debug_stats_merge(v.ctypes.data_as(ctypes.c_void_p), n, chunk, out.ctypes.data_as(ctypes.c_void_p))
        v64 = v.astype(np.float64)
        assert out[0] == n
        assert abs(out[1] - v64.mean()) <= 1e-6 * max(1.0, abs(v64.mean())), (n, out[1], v64.mean())
        var = ((v64 - v64.mean()) ** 2).sum() / n
        assert abs(out[2] / n - var) <= 2e-5 * var, (n, mean, out[2] / n, var)

"""GPU parity against REFERENCE output directly (not through the C oracle): the HIP FPS / ball-query / Chamfer kernels
vs goldens produced by the reference's own pure-torch helpers (tests/golden/make_golden_pins.py):
farthest_point_sample (models/encoders/pointnet2_utils.py:60-81, start index patched to 0), query_ball_point (:84-104),
distChamfer (datasets/evaluation_utils.py:93-103), the feature-propagation path's 3-NN + interpolation (:289-299).  Only the
auction EMD has no pure-torch counterpart in the reference and stays "parity unpinned" (bit-exact vs oracle/pointnet2.c only)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_pointnet2_cpu import (BALLQ_PINS, CHAMFER_ATOL, CHAMFER_PINS, FPS_PINS, GOLDEN, THREE_NN_PINS,  # noqa: E402
                                        check_ballquery_pin, check_three_nn_pin)


@pytest.fixture(scope="module")
def pu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.pointnet2_ops import pointnet2_utils
    return pointnet2_utils


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", FPS_PINS)
def test_fps_matches_reference_torch_fps(pu, name):
    g = np.load(os.path.join(GOLDEN, name))
    got = pu.furthest_point_sample(dev(g["xyz"]), int(g["npoint"])).cpu().numpy()
    assert np.array_equal(got, g["idx"])


@pytest.mark.parametrize("name", BALLQ_PINS + ["pn2_torch_ballquery.npz"])
def test_ball_query_matches_reference_torch_helper(pu, name):
    g = np.load(os.path.join(GOLDEN, name))
    got = pu.ball_query(float(g["radius"]), int(g["nsample"]), dev(g["xyz"]), dev(g["new_xyz"])).cpu().numpy()
    check_ballquery_pin(got, g)


@pytest.mark.parametrize("name", CHAMFER_PINS)
def test_chamfer_matches_reference_distChamfer(pu, name):
    from difffacto_amd.metrics import ChamferFunction
    g = np.load(os.path.join(GOLDEN, name))
    d1, d2 = ChamferFunction.apply(dev(g["a"]), dev(g["b"]))
    np.testing.assert_allclose(d1.cpu().numpy(), g["dist_a"], rtol=0, atol=CHAMFER_ATOL)
    np.testing.assert_allclose(d2.cpu().numpy(), g["dist_b"], rtol=0, atol=CHAMFER_ATOL)


@pytest.mark.parametrize("name", THREE_NN_PINS)
def test_three_nn_and_interpolate_match_the_reference_torch_path(pu, name):
    g = np.load(os.path.join(GOLDEN, name))
    dist, idx = pu.three_nn(dev(g["unknown"]), dev(g["known"]))            # (sqrt(dist2), idx) like P2U:124-125
    interp = pu.three_interpolate(dev(g["feats"]), dev(g["idx"]), dev(g["weight"]))
    check_three_nn_pin(dist.cpu().numpy() ** 2, idx.cpu().numpy(), interp.cpu().numpy(), g)

"""Random-shape sweep of the GPU parity tests: python tools/fuzz_parity.py [seconds] [seed].

Calls the parametrised oracle comparisons of tests/test_gpu_*.py (the same assertions, the same tolerances) with shapes
drawn at random instead of the fixed lists, until the time budget is spent; prints one line per family with the number of
cases and every failing argument tuple.  Exit code 1 if anything failed.  The oracle is the checker here, as in the tests.
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ri = lambda lo, hi: int(rng.integers(lo, hi + 1))
rb = lambda: bool(rng.integers(0, 2))

import test_gpu_denoiser as td  # noqa: E402
import test_gpu_emd as te  # noqa: E402
import test_gpu_encoder_train as tet  # noqa: E402
import test_gpu_pointnet2 as tp  # noqa: E402
import test_gpu_train as tt  # noqa: E402
from difffacto_amd import synth  # noqa: E402
from difffacto_amd.pointnet2_ops import pointnet2_utils as pu  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pnv2_conditioning import check_case as pnv2_check  # noqa: E402

W = synth.make_denoiser_weights(seed=0)


def unwrap(f):
    return getattr(f, "__wrapped__", f)


def gemm_case():
    tn = ri(0, 1)
    if tn == 0:   # NT: N % 128 == 0, K % 32 == 0, M >= 256
        return (0, ri(256, 3000), 128 * ri(1, 8), 32 * ri(1, 32), ri(0, 1), 0)
    return (1, 128 * ri(1, 8), 128 * ri(1, 4), ri(33, 9000), ri(0, 1), ri(0, 1))


FAMILIES = [
    # name, callable, argument generator, rough cost weight
    ("denoiser eps f32 vs oracle", lambda a: unwrap(td.test_eps_f32_vs_oracle_seeded)(W, *a), lambda: (ri(1, 6), 32 * ri(1, 50), rb())),
    ("denoiser train fwd/bwd vs oracle", lambda a: tt.test_forward_backward_vs_oracle_full_gradients(*a),
     lambda: (ri(1, 6), 32 * ri(1, 24), rb(), rb())),      # the training path takes N % 32 == 0 (include/dfx.h)
    ("bf16 product kernels", lambda a: tt.test_bf16_product_kernels_against_torch(*a), gemm_case),
    # (against the float64 oracle with the fp32 oracle's own error as the yardstick: tools/pnv2_conditioning.py says why)
    ("PointNetV2 train vs f64 oracle", lambda a: pnv2_check(*a), lambda: (ri(2, 8), ri(64, 800))),
    ("prior loss vs oracle", lambda a: tet.test_prior_loss_vs_oracle_other_batch(*a), lambda: (ri(2, 90), False)),
    ("FPS", lambda a: tp.test_fps_matches_oracle(pu, *a), lambda: (lambda n: (ri(1, 4), n, ri(1, min(n, 700))))(ri(1, 6000))),
    ("ball query", lambda a: tp.test_ball_query_matches_oracle(pu, *a),
     lambda: (lambda n: (ri(1, 3), n, ri(1, min(n, 300)), float(rng.uniform(0.05, 0.6)), ri(1, 64)))(ri(1, 4000))),
    ("gather", lambda a: tp.test_gather_matches_oracle(pu, *a), lambda: (ri(1, 4), ri(1, 140), ri(1, 3000), ri(1, 3000))),
    ("group", lambda a: tp.test_group_matches_oracle(pu, *a), lambda: (ri(1, 3), ri(1, 140), ri(1, 2500), ri(1, 300), ri(1, 64))),
    ("three_nn / interpolate", lambda a: tp.test_three_nn_interpolate_match_oracle(pu, *a), lambda: (ri(1, 3), ri(1, 2500), ri(3, 3000), ri(1, 20))),
    ("chamfer", lambda a: tp.test_chamfer_matches_oracle(pu, *a), lambda: (ri(1, 3), ri(1, 2500), ri(1, 2500))),
    ("EMD auction", lambda a: te.test_emd_forward_bit_exact_vs_oracle(*a), lambda: (ri(1, 3), ri(8, 1200), ri(1, 80))),
]

for name in ("test_fps_matches_oracle", "test_ball_query_matches_oracle"):
    assert hasattr(tp, name), name

t_end = time.time() + budget
stats = {n: [0, []] for n, _, _ in [(f[0], 0, 0) for f in FAMILIES]}
k = 0
while time.time() < t_end:
    name, fn, gen = FAMILIES[k % len(FAMILIES)][:3]
    k += 1
    args = gen()
    try:
        fn(args)
        stats[name][0] += 1
    except Exception as e:  # noqa: BLE001
        stats[name][0] += 1
        stats[name][1].append((args, (str(e) or traceback.format_exc())[:300].replace("\n", " ")))
bad = 0
for name, (n, fails) in stats.items():
    print(f"{name:36s} {n:4d} cases, {len(fails)} failed")
    for a, msg in fails:
        bad += 1
        print(f"    FAIL {a}: {msg}")
sys.exit(1 if bad else 0)

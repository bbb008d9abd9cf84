"""Random-shape sweep of the GPU parity tests: python tools/fuzz_parity.py [seconds] [seed] [family filter].

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
only = sys.argv[3] if len(sys.argv) > 3 else ""      # substring of the family names to run
rng = np.random.default_rng(seed)
ri = lambda lo, hi: int(rng.integers(lo, hi + 1))
rb = lambda: bool(rng.integers(0, 2))

import test_gpu_denoiser as td  # noqa: E402
import test_gpu_emd as te  # noqa: E402
import test_gpu_encoder_train as tet  # noqa: E402
import test_gpu_latents as tl  # noqa: E402
import test_gpu_pointnet2 as tp  # noqa: E402
import test_gpu_train as tt  # noqa: E402
from difffacto_amd import synth  # noqa: E402
from difffacto_amd.pointnet2_ops import pointnet2_utils as pu  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
from pnv2_conditioning import check_case as pnv2_check  # noqa: E402

W = synth.make_denoiser_weights(seed=0)


def unwrap(f):
    return getattr(f, "__wrapped__", f)


def gemm_case():
    tn = ri(0, 1)
    if tn == 0:   # NT: N % 128 == 0, K % 32 == 0, M >= 256
        return (0, ri(256, 3000), 128 * ri(1, 8), 32 * ri(1, 32), ri(0, 1), 0)
    return (1, 128 * ri(1, 8), 128 * ri(1, 4), ri(33, 9000), ri(0, 1), ri(0, 1))


def chain_case(T, B, N, all_valid, interval, seed):
    """DDPM chain in one launch, explicit noise, trajectory snapshots: tests/test_gpu_denoiser.py::test_chain_f32_vs_oracle_T100
    at other sizes."""
    import torch
    from oracle import diffusion as odf
    eng = td._engine(W, T, "f32")
    part_code, mean, logvar, valid = synth.make_latents(B, seed=seed, all_valid=all_valid)
    seg = synth.make_seg_mask(valid, N)
    var = np.exp(logvar).astype(np.float32)
    r = np.random.default_rng(seed)
    xT = r.standard_normal((B, 3, N)).astype(np.float32)
    zs = r.standard_normal((T, B, 3, N)).astype(np.float32)
    anchors, variance = odf.gather_params(seg, mean, var)
    dec = odf.decode(odf.Tables(T), W, anchors, [part_code, np.concatenate([mean, var], 1)], variance, seg, valid, xT, zs,
                     ret_traj=True, ret_interval=interval)
    ctx = eng.prepare_shapes(*map(torch.from_numpy, (part_code, mean, var, valid)))
    pred, traj = eng.sample_chain(ctx, torch.from_numpy(seg), x_T_noise=torch.from_numpy(xT), step_noise=torch.from_numpy(zs),
                                  ret_interval=interval)
    assert np.abs(pred.cpu().numpy() - dec["pred"]).max() < td.TOL_F32_CHAIN
    for k, t in enumerate(eng.snapshot_times(interval)):
        assert np.abs(traj[k].cpu().numpy() - dec[t]).max() < td.TOL_F32_CHAIN, t


LW = synth.make_latent_weights(seed=0)
_sampler = []


def latents_case(S, K, N, fixed, seed):
    """flows in reverse + part aligner + sample_latents glue: tests/test_gpu_latents.py::test_sample_latents_vs_oracle_full_batch
    at other sizes and fixed-part patterns."""
    import torch
    from difffacto_amd.latents import LatentSampler
    from oracle import latents as ol
    if not _sampler:
        _sampler.append(LatentSampler(LW, noise_scale=100.0))
    r = np.random.Generator(np.random.PCG64(seed))
    w = r.standard_normal((S, 256, 4)).astype(np.float32)
    an = r.standard_normal((S * K, 32)).astype(np.float32)
    _, _, _, valid = synth.make_latents(S, seed=seed)
    ref = ol.sample_latents(LW, w, an, valid, fixed, K, N, noise_scale=100.0)
    cu = lambda a: torch.from_numpy(a).cuda()
    out = _sampler[0].sample_latents(cu(w), cu(an), cu(valid), fixed_id=fixed, K=K, npoints=N)
    assert np.array_equal(out["seg_mask"].cpu().numpy(), ref["seg_mask"])
    assert np.array_equal(out["valid_id"].cpu().numpy(), ref["valid_id"])
    for k in ("part_code", "mean", "logvar", "mean_per_point", "logvar_per_point"):
        tl._close(out[k], ref[k])


FAMILIES = [
    ("DDPM chain f32 vs oracle", lambda a: chain_case(*a), lambda: (ri(2, 40), ri(1, 4), 32 * ri(1, 12), rb(), ri(1, 12), ri(0, 10 ** 6))),
    ("bf16 pipelined partial tiles = full tiles", lambda a: td.test_pipelined_kernel_partial_tiles_are_bit_identical_to_full_tiles(W, *a),
     lambda: (32 * ri(3, 16),)),
    ("sample_latents vs oracle", lambda a: latents_case(*a),
     lambda: (ri(1, 70), ri(1, 4), 4 * ri(8, 128), [ri(0, 1) for _ in range(4)], ri(0, 10 ** 6))),
    # name, callable, argument generator, rough cost weight
    ("denoiser eps f32 vs oracle", lambda a: unwrap(td.test_eps_f32_vs_oracle_seeded)(W, *a), lambda: (ri(1, 6), 32 * ri(1, 50), rb())),
    ("denoiser train fwd/bwd vs oracle", lambda a: tt.test_forward_backward_vs_oracle_full_gradients(*a),
     lambda: (ri(1, 6), 32 * ri(1, 24), rb(), rb())),      # the training path takes N % 32 == 0 (include/dfx.h)
    ("bf16 product kernels", lambda a: tt.test_bf16_product_kernels_against_torch(*a), gemm_case),
    # (against the float64 oracle with the fp32 oracle's own error as the yardstick: tools/experiments/pnv2_conditioning.py says why)
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
    # round 2: the bf16 chain kernels' workgroup-size variants and the co-operative latency kernel against each other (bit for bit),
    # and the fused training feed-forward against the layer-by-layer kernels
    ("bf16 kernels 8/4/2-wave + co-operative: bit-identical", lambda a: td.test_small_batch_workgroup_sizes_are_bit_identical(W, *a),
     lambda: (32 * ri(3, 100),)),
    # round 3: the exact-fp32 persistent chain (k_denoise_pipe_f32, 8 / 4 / 2 wavefronts) against the direct fp32 kernel, bit for bit;
    # the small-batch family above now includes k_denoise_pipe2 (two point tiles per wavefront) as variant 64
    ("f32 pipelined = direct fp32 kernel: bit-identical", lambda a: td.test_f32_pipelined_chain_is_bit_identical_to_the_direct_kernel(W, *a),
     lambda: (32 * ri(3, 100),)),
    ("fused vs layer-by-layer training FF", lambda a: tt.test_fused_feed_forward_matches_the_layer_by_layer_bf16_path(*a, **(dict(g_max=4e-2, g_l2=3e-2) if a[2] else {})),
     lambda: (lambda n: (max(1, -(-256 // n)) + ri(0, 3), n, (0.2, ri(1, 10 ** 6)) if rb() else None))(32 * ri(1, 40))),   # (B, N, dropout: round 5)
    # round 6: config 5 as shipped (bf16 products + Dropout 0.2) against the fp32 autograd oracle DIRECTLY, Philox factors replayed at the reference's
    # sites; both bf16 paths (fused kernels / layer-by-layer), the path asserted from the library's record
    ("bf16 + dropout vs oracle (replayed factors)", lambda a: tt.test_bf16_dropout_paths_against_the_oracle_with_replayed_factors(*a, g_max=5e-2, g_l2=4e-2),
     lambda: (lambda n: (max(1, -(-256 // n)) + ri(0, 2), n, "fused" if rb() else "layer"))(32 * ri(1, 48))),
]

for name in ("test_fps_matches_oracle", "test_ball_query_matches_oracle"):
    assert hasattr(tp, name), name

FAMILIES = [f for f in FAMILIES if only in f[0]]
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

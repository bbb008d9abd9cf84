"""GPU parity gates on the HEADLINE configuration (BASELINE.json configs[1]: gen_chair, 2048 points x 4 parts, T = 1000 DDPM
steps, bf16 `k_denoise_pipe`) and on the multi-GPU contract of SURVEY.md §8(e) (results independent of the GPU count;
`bench.py --gpus 2` executed as two ranks on the one GPU of the box).

Weight set: with random-init weights eps_theta is an arbitrary O(1) function of x and the T = 1000 chain is chaotic (clouds blow
up to +-20): parity numbers on it say little.  `contractive_weights` scales `proj_out` by 0.05, so eps_theta ~ 0 and
the reverse chain is the linear map (x - a) -> (x - a) / sqrt(alpha_t) per step (anchored_diffusion.py:175-193 with eps = 0):
deviations grow by at most 1 / sqrt(alphas_cumprod[T-1]) = 157 like the cloud itself, so errors RELATIVE to the cloud extent are
meaningful.  Every kernel path (all GEMMs, LayerNorms, softmax, GELU, posterior) runs exactly as in the headline bench."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # tests/_variants.py

pytestmark = pytest.mark.gpu

from difffacto_amd import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# measured on MI355X (profiles/r04_parity_prints.txt; r02 / r03 values in brackets: before b1' moved onto the constant-one K slot of
# GEMM1); every gate is <= 3x the measured value:
#   bf16_f32   max-abs / cloud extent, bf16 pipelined kernel vs exact-fp32 chain            (measured 1.4e-5 [1.2e-5] | see below)
#   f32_oracle exact-fp32 HIP chain vs the PyTorch-CPU oracle on a 256-point subset / extent (measured 4.8e-7)
#   cd         Chamfer-L2(bf16, fp32) / extent^2                                             (measured 8.5e-11 [6.0e-11])
#   emd        auction EMD(bf16, fp32) on the unit-box-normalised clouds                     (measured 6.1e-6 [5.0e-6])
GATES = {"contractive": dict(bf16_f32=3.6e-5, f32_oracle=5e-6,   # f32_oracle: ~10x measured (CPU BLAS summation order varies by host)
                                cd=1.8e-10, emd=1.5e-5),
         # the random-init set is chaotic over 1000 steps (no oracle leg: fp32 rounding differences between two fp32
         # implementations grow the same way); it is kept as the worst case for the bf16 deviation
         "random-init": dict(bf16_f32=9e-4, f32_oracle=None, cd=6.5e-8, emd=2.8e-4)}   # measured 4.2e-4, 2.9e-8, 1.08e-4 [r03: 3.1e-4, 2.2e-8, 9.3e-5]


def contractive_weights():
    W = synth.make_denoiser_weights(seed=0)
    W["proj_out.weight"] = (W["proj_out.weight"] * 0.05).astype(np.float32)
    W["proj_out.bias"] = (W["proj_out.bias"] * 0.05).astype(np.float32)
    return W


def _engine(W, T, prec):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.engine import DenoiserEngine
    return DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision=prec)


@pytest.mark.parametrize("wset,nw", [("contractive", 8), ("random-init", 8), ("contractive", 0)])
def test_headline_T1000_N2048_bf16_pipe_vs_f32_and_oracle(wset, nw):
    """nw = 8: the kernels bench.py times at B = 128 — `k_denoise_pipe<8>` (bf16) and `k_denoise_pipe_f32<8>` (fp32) — FORCED
    (at B = 8 the launcher would otherwise take the co-operative kernel / two-wavefront workgroups) and asserted through
    dfx_last_kernel_variant; nw = 0 keeps the automatic choice covered."""
    from _variants import forced, ran
    from difffacto_amd.metrics import EMD, chamfer_l2
    from oracle import diffusion as odf
    from oracle import torch_cpu as tc
    T, B, N = 1000, 8, 2048
    W = contractive_weights() if wset == "contractive" else synth.make_denoiser_weights(seed=0)
    gate = GATES[wset]
    pc, mean, logvar, valid = synth.make_latents(B, seed=5)
    var = np.exp(logvar).astype(np.float32)
    seg = synth.make_seg_mask(valid, N)
    g = torch.Generator(device="cuda").manual_seed(1)
    xT = torch.randn(B, 3, N, device="cuda", generator=g)
    zs = torch.randn(T, B, 3, N, device="cuda", generator=g)
    args = tuple(map(torch.from_numpy, (pc, mean, var, valid)))
    out = {}
    took = {}
    for prec in ("f32", "bf16"):
        eng = _engine(W, T, prec)
        with forced(nw):
            out[prec], _ = eng.sample_chain(eng.prepare_shapes(*args), torch.from_numpy(seg), x_T_noise=xT, step_noise=zs)
            took[prec] = ran(prec, nw)
        eng.close()
    print(f"headline[{wset}] kernels: bf16 -> {took['bf16']}, f32 -> {took['f32']}")
    assert torch.isfinite(out["bf16"]).all() and torch.isfinite(out["f32"]).all()
    extent = float((out["f32"].amax((1, 2)) - out["f32"].amin((1, 2))).mean())
    rel = float((out["bf16"] - out["f32"]).abs().max()) / extent
    cd = float(chamfer_l2(out["bf16"], out["f32"]).mean()) / extent ** 2
    lo = torch.minimum(out["bf16"].amin((1, 2), keepdim=True), out["f32"].amin((1, 2), keepdim=True))
    hi = torch.maximum(out["bf16"].amax((1, 2), keepdim=True), out["f32"].amax((1, 2), keepdim=True))
    emd = float(EMD(0.002, 10000, True)(((out["bf16"] - lo) / (hi - lo)).contiguous(), ((out["f32"] - lo) / (hi - lo)).contiguous()).mean())

    print(f"headline[{wset}] T={T} B={B} N={N}: extent {extent:.2f}; bf16 vs f32 max-abs/extent {rel:.3e}, Chamfer-L2/extent^2 {cd:.3e}, "
          f"EMD(unit box) {emd:.3e}")
    assert rel < gate["bf16_f32"] and cd < gate["cd"] and emd < gate["emd"]
    if gate["f32_oracle"] is None:
        return
    # the oracle on a subset: points are independent given the shape's 4 part tokens, so 256 points of shape 0 with their
    # own noise columns reproduce those points of the full run (PyTorch-CPU restatement, pinned to the reference goldens)
    sub = np.arange(0, N, N // 256)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    anchors, variance = odf.gather_params(seg[:1, sub], mean[:1], var[:1])
    ctx = [tt(pc[:1]), tt(np.concatenate([mean[:1], var[:1]], 1))]
    Wt = {k: tt(v) for k, v in W.items()}
    tb = odf.Tables(T)
    xTs, zss = xT[:1, :, sub].cpu(), zs[:, :1][:, :, :, sub].cpu()
    with torch.no_grad():
        x = torch.sqrt(tt(variance)) * xTs + tt(anchors)
        for i, t in enumerate(range(T - 1, -1, -1)):
            x, _ = tc.p_sample(tb, Wt, x, t, tt(anchors), ctx, tt(variance), tt(seg[:1, sub]), tt(valid[:1]), zss[i])
    ref = x.transpose(1, 2)[0]
    rel_or = float((out["f32"][0, sub].cpu() - ref).abs().max()) / extent
    rel_or_bf16 = float((out["bf16"][0, sub].cpu() - ref).abs().max()) / extent
    print(f"headline[{wset}]: f32 vs oracle (256 pts) / extent {rel_or:.3e}; bf16 vs oracle {rel_or_bf16:.3e}")
    assert rel_or < gate["f32_oracle"]
    assert rel_or_bf16 < gate["bf16_f32"]


@pytest.mark.parametrize("prec,N", [("bf16", 2048), ("f32", 256)])
def test_philox_noise_is_keyed_by_the_global_shape_id(prec, N):
    """one call over 6 shapes == two calls over 4 + 2 shapes with shape_offset, bit for bit (in-kernel
    Philox, both the persistent pipelined kernel and the direct one)."""
    T, B = 5, 6
    W = synth.make_denoiser_weights(seed=0)
    eng = _engine(W, T, prec)
    pc, mean, logvar, valid = synth.make_latents(B, seed=3)
    var = np.exp(logvar).astype(np.float32)
    seg = torch.from_numpy(synth.make_seg_mask(valid, N))
    t = lambda a, lo, hi: torch.from_numpy(np.ascontiguousarray(a[lo:hi]))
    full, _ = eng.sample_chain(eng.prepare_shapes(*(t(a, 0, B) for a in (pc, mean, var, valid))), seg, seed=11)
    parts = []
    for lo, hi in ((0, 4), (4, 6)):
        c = eng.prepare_shapes(*(t(a, lo, hi) for a in (pc, mean, var, valid)))
        parts.append(eng.sample_chain(c, seg[lo:hi], seed=11, shape_offset=lo)[0])
    assert torch.equal(full, torch.cat(parts))
    wrong = eng.sample_chain(eng.prepare_shapes(*(t(a, 4, 6) for a in (pc, mean, var, valid))), seg[4:6], seed=11)[0]
    assert not torch.equal(full[4:], wrong)   # without the offset the second block would repeat the first block's stream


def _run_bench(nproc, extra, env_extra, dump):
    env = dict(os.environ, **env_extra)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    base = ["--steps", "1", "--warmup", "0", "--timesteps", "20", "--no-cpu-baseline", "--no-train-line", "--no-parity",
            "--dump-clouds", dump] + extra
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(29611 + nproc), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + base
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_on_one_gpu_gloo_and_gpu_count_independence(tmp_path):
    """the N > 1 path of bench.py (process-group init, weight broadcast, sharded chain with shape_offset,
    gather of the clouds, max-over-ranks timing) executed on hardware as two ranks sharing the box's GPU (gloo backend, device
    tensors staged through the host), and the same job (8 shapes) on one rank: identical clouds."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    two = _run_bench(2, ["--batch", "4"], {"DFX_BENCH_BACKEND": "gloo"}, str(tmp_path / "two.npy"))
    assert two["n_gpus"] == 2 and two["config"]["shapes_gathered"] == 8 and np.isfinite(two["value"]) and two["value"] > 0
    assert two["config"]["weights_bcast_ms"] > 0 and two["scaling"] == "weak"
    one = _run_bench(1, ["--batch", "8"], {}, str(tmp_path / "one.npy"))
    assert one["n_gpus"] == 1 and one["config"]["shapes_gathered"] == 8
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == b.shape == (8, 2048, 3)
    assert np.array_equal(a, b)


def test_bench_eight_ranks_on_one_gpu_and_strong_scaling(tmp_path):
    """VERDICT r2 item 5 (a, b, d): bench.py --gpus 8 --batch 2 as EIGHT gloo ranks sharing the box's GPU (the driver's 8-GPU launch
    line, collective library swapped): the 16 clouds equal the one-rank 16-shape job bit for bit, and the JSON carries what the
    process group itself reports.  Then BASELINE configs[3]'s mode in miniature: --total-shapes 10 over four ranks (ragged 3, 3, 2, 2)
    through the DFX_GATHER=all_gather fallback = the one-rank 10-shape job."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    eight = _run_bench(8, ["--batch", "2"], {"DFX_BENCH_BACKEND": "gloo"}, str(tmp_path / "eight.npy"))
    c = eight["config"]
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak" and c["shapes_gathered"] == 16 and c["total_shapes"] == 16
    assert c["backend"] == "gloo" and c["world_size_seen"] == 8 and len(c["rank_devices"]) == 8 and c["shapes_per_rank"] == [2] * 8
    assert c["weights_bcast_bytes"] > 10_000_000 and c["weights_bcast_ms"] > 0 and c["gather"] == "gather"
    one = _run_bench(1, ["--batch", "16"], {}, str(tmp_path / "one.npy"))
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "eight.npy")
    assert a.shape == b.shape == (16, 2048, 3) and np.array_equal(a, b)
    four = _run_bench(4, ["--total-shapes", "10"], {"DFX_BENCH_BACKEND": "gloo", "DFX_GATHER": "all_gather"}, str(tmp_path / "four.npy"))
    c = four["config"]
    assert four["scaling"] == "strong" and c["shapes_per_rank"] == [3, 3, 2, 2] and c["gather"] == "all_gather" and c["total_shapes"] == 10
    ten = _run_bench(1, ["--batch", "10"], {}, str(tmp_path / "ten.npy"))
    assert ten["scaling"] == "weak"
    assert np.array_equal(np.load(tmp_path / "ten.npy"), np.load(tmp_path / "four.npy"))


def test_bench_two_ranks_rccl():
    """The same through RCCL ("nccl"), one rank per GPU: only on boxes with >= 2 GPUs (the driver's scaling node)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        two = _run_bench(2, ["--batch", "4"], {}, os.path.join(d, "two.npy"))
        one = _run_bench(1, ["--batch", "8"], {}, os.path.join(d, "one.npy"))
        assert two["n_gpus"] == 2 and np.array_equal(np.load(os.path.join(d, "one.npy")), np.load(os.path.join(d, "two.npy")))


def test_bench_line_contract_single_gpu(tmp_path):
    """The N = 1 JSON line of bench.py (driver contract + the tier's `roofline` / `cpu_baseline` objects + round 2's `parity` and
    `t100` blocks) on a reduced run: every key present, numbers finite, the workload named."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--timesteps", "20", "--batch", "32",
                        "--no-train-line"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "t100", "small_batch", "f32", "sweep"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "shapes/s" and d["dtype"] == "bf16" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert "error" not in d["parity"] and d["parity"]["bf16"]["max_abs"] < 3e-3 and d["parity"]["f32"]["max_abs"] < 1e-4   # measured 9.2e-4 / 2.4e-6
    assert "error" not in d["t100"] and d["t100"]["shapes_per_s"] > d["t100"]["wall_shapes_per_s"] > 0
    assert "error" not in d["f32"] and d["f32"]["dtype"] == "f32" and d["f32"]["finite"] and 0 < d["f32"]["frac"] < 1
    assert sorted(d["sweep"]) == ["gen_airplane", "gen_car", "gen_chair_B1024", "gen_lamp", "outlier_weights"]
    ow = d["sweep"]["outlier_weights"]   # round 6: a LayerNorm3 gain outlier in channel 127 keeps the fold (on another channel) and the headline kernel
    assert ow["kernel_variant"] == "k_denoise_pipe<8>" and ow["w1_fold"]["folded"] and ow["w1_fold"]["channel"] not in (127, -1), ow
    assert all("error" not in v and v["finite"] and v["shapes_per_s"] > 0 for v in d["sweep"].values()) and d["sweep"]["gen_car"]["npoints"] == 8192
    assert "error" not in d["small_batch"] and d["small_batch"]["B1"]["ms_per_chain"] > 0 and d["small_batch"]["B4"]["shapes_per_s"] > 0


def test_data_parallel_training_example_two_ranks_on_one_gpu():
    """the documented multi-GPU training launch (examples/train_denoiser.py under torch.distributed.run) executed as two
    ranks sharing the box's GPU (gloo): parameter broadcast of leaf nn.Parameters, one flat gradient all-reduce per iteration,
    clip + Adam on identical gradients; rank 0 reports every iteration."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, DFX_BENCH_BACKEND="gloo")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(ROOT, "examples", "train_denoiser.py"), "--iters", "3", "--batch", "4", "--npoints", "256", "--timesteps", "100"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("iter ")]
    assert len(lines) >= 3 and all("mse_loss" in l and "grad norm" in l for l in lines), r.stdout[-1500:]

#!/usr/bin/env python
"""Headline benchmark: generated shapes/sec for gen_chair (2048 pts x 4 parts, 1000-step DDPM).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch (anchor_gen.py:1034-1084): the latent sampler
(torch.randn draws -> flows in reverse -> part aligner -> seg ids; dfx_sample_latents), per-batch
shape-context preparation, and the full T-step reverse chain (persistent kernel, in-kernel Philox
noise) for B shapes per GPU, plus — for N>1 — the gather of the generated clouds to rank 0.
Part-presence patterns (valid_id) are synthetic and already resident in HBM when the timed region starts.  Weak scaling: B shapes per GPU, independent shapes,
no per-step collective; the frozen weights are broadcast from rank 0 once (RCCL), outside the timed
region (reported as weights_bcast_ms).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL's intra-node transport needs this before the HIP runtime starts (the driver's
# environment exports it already; a bare `python -m torch.distributed.run bench.py` gets it here)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from difffacto_amd import synth  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md (dense MFMA)


def flops_per_step(N, depth=5):
    """Algorithmic matmul FLOPs of one denoiser evaluation for one shape."""
    per_point = 13 * 128 + depth * (128 * 128 + 128 * 128 + 128 * 1024 + 512 * 128 + 2 * 8 * 4 * 16) + 128 * 3
    return 2 * (N * per_point + depth * 2 * 4 * 522 * 128 + 256 * 2048 + 1024 * 256)


def rccl_transport_summary():
    """Version banner + distinct channel transports from this process's NCCL_DEBUG_FILE (see main()), or None."""
    import re
    path = os.environ.get("NCCL_DEBUG_FILE", "")
    if not path or not os.path.exists(path):
        return None
    try:
        version, via = None, {}
        for line in open(path, errors="replace"):
            if version is None and ("RCCL version" in line or "NCCL version" in line):
                version = line.strip()[-120:]
            m = re.search(r"via (\S+)", line)
            if m:
                via[m.group(1)] = via.get(m.group(1), 0) + 1
        return {"banner": version, "channels_via": via}
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:120]}


def _newest_profile(pattern):
    """Newest committed profiles/rNN_<pattern> (by round number), or None."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)):
        r = int(re.match(r"r(\d+)_", os.path.basename(f)).group(1))
        if best is None or r > best[0]:
            best = (r, f)
    return best


def power_limited_ceiling():
    """What the chip sustains on bare MFMA streams with random bf16 operands at the 1.4 kW cap, from the newest committed run of
    tools/ubench/pair_issue.hip (full-chip rows, TFLOP/s of EXECUTED MFMAs): bare MFMAs, + one LDS fragment refill per MFMA, + the
    kernel's own mix (refill per MFMA + 5 packed VALU per MFMA).  Context for `frac`, not a replacement for the nominal peak."""
    import re
    best = _newest_profile("ubench_pair_issue.txt")
    if best is None:
        return None
    rows = {}
    for line in open(best[1]):
        m = re.match(r"(.+?)\s+blocks\s+(\d+):.*-> (\d+) TFLOP/s", line)
        if m and int(m.group(2)) > 1:
            rows[m.group(1).strip()] = float(m.group(3))
    pick = lambda k: rows.get(k)
    return {"source": os.path.relpath(best[1], ROOT), "round": f"r{best[0]:02d}",
            "bare_mfma_tflops": pick("C/D VGPR, B VGPR, no refill, no filler"),
            "refill_per_mfma_tflops": pick("C/D AGPR, refill per MFMA (own A)"),
            "kernel_mix_tflops": pick("C/D AGPR, refill per MFMA, 5 v_pk per MFMA")}


class BoxSampler:
    """Clock / power samples of the GPU while the timed region runs (a host thread reading the amdgpu hwmon files every 25 ms; `rocm-smi --json`
    snapshots when those are not readable): the bench line then says at which clock and power its number was measured — boxes of the pool
    differ, and the chain kernel runs at the power cap.  Reading sysfs costs the GPU nothing."""

    def __init__(self, device_index=0, period_s=0.025):
        import glob
        import threading
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        self.files = {}
        cards = []
        try:   # the hwmon directory of THIS HIP device (a box exposes all eight cards in sysfs, one of them to the process): by PCI address
            pr = torch.cuda.get_device_properties(device_index)
            cards = glob.glob("/sys/bus/pci/devices/%04x:%02x:%02x.0/hwmon/hwmon*" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
        except Exception:   # noqa: BLE001
            pass
        if not cards:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        cards = [c for c in cards if any(os.path.exists(os.path.join(c, f)) for f in ("power1_average", "power1_input"))]
        if cards:
            h = cards[0] if len(cards) == 1 else cards[min(device_index, len(cards) - 1)]
            for key, names in (("power_uw", ("power1_average", "power1_input")), ("sclk_hz", ("freq1_input",)), ("temp_mc", ("temp2_input", "temp1_input"))):
                for n in names:
                    if os.path.exists(os.path.join(h, n)):
                        self.files[key] = os.path.join(h, n)
                        break
            self.source = h
        else:
            self.source = "rocm-smi"
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        if self.files:
            out = {}
            for k, f in self.files.items():
                try:
                    out[k] = float(open(f).read().split()[0])
                except Exception:   # noqa: BLE001
                    pass
            return out
        import subprocess
        try:
            j = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)
            card = j[sorted(j)[0]]
            out = {}
            for k, v in card.items():
                lk = k.lower()
                if "power" in lk and "w" in lk:
                    try:
                        out["power_uw"] = float(v) * 1e6
                    except Exception:   # noqa: BLE001
                        pass
                if lk.startswith("sclk clock speed"):
                    try:
                        out["sclk_hz"] = float(str(v).strip("()").lower().replace("mhz", "")) * 1e6
                    except Exception:   # noqa: BLE001
                        pass
            return out
        except Exception:   # noqa: BLE001
            return {}

    def _run(self):
        while not self._stop.is_set():
            s = self._read()
            if s:
                s["t"] = time.perf_counter()
                self.samples.append(s)
            self._stop.wait(self.period)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self, t0=None, t1=None):
        rows = [s for s in self.samples if (t0 is None or s["t"] >= t0) and (t1 is None or s["t"] <= t1)]
        out = {"source": self.source, "samples": len(rows)}
        for key, name, scale in (("power_uw", "power_w", 1e-6), ("sclk_hz", "sclk_mhz", 1e-6), ("temp_mc", "temp_c", 1e-3)):
            v = [r[key] * scale for r in rows if key in r]
            if v:
                out[name] = {"min": min(v), "mean": float(np.mean(v)), "max": max(v)}
        return out


def box_calibration(target_ms=60.0):
    """`roofline.box_bare_mfma_tflops`: what THIS chip sustains on a bare v_mfma_f32_32x32x16_bf16 stream (one wavefront per SIMD on every CU,
    pseudo-random operands; dfx_debug_bare_mfma = tools/ubench/pair_issue.hip's first row inside the library), measured in this process right
    behind the timed region — the same box, the same thermal state.  `frac_of_box` = the chain kernel's EXECUTED matrix rate over it is the
    round-over-round comparable figure; `frac` (algorithmic FLOPs over the nominal 2.5 PFLOP/s) stays the headline roofline number."""
    import ctypes
    from difffacto_amd import _ffi
    try:
        ms, tf = ctypes.c_float(0), ctypes.c_double(0)
        _ffi.check(_ffi.lib().dfx_debug_bare_mfma(20000, ctypes.byref(ms), ctypes.byref(tf), _ffi.current_stream()), "dfx_debug_bare_mfma")
        iters = max(20000, int(20000 * target_ms / max(ms.value, 1e-3)))
        runs = []
        for _ in range(3):
            _ffi.check(_ffi.lib().dfx_debug_bare_mfma(iters, ctypes.byref(ms), ctypes.byref(tf), _ffi.current_stream()), "dfx_debug_bare_mfma")
            runs.append((float(ms.value), float(tf.value)))
        return {"bare_mfma_tflops": float(np.median([r[1] for r in runs])), "runs_ms_tflops": runs, "iters": iters,
                "what": "bare v_mfma_f32_32x32x16_bf16 stream, 1 wavefront per SIMD on every CU, pseudo-random bf16 operands (dfx_debug_bare_mfma); "
                        "executed MFMA TFLOP/s of this chip right behind the timed region"}
    except Exception as e:   # noqa: BLE001 - a calibration failure must not cost the headline line
        return {"error": repr(e)[:200]}


def fps_line(dev):
    """gen_car's evaluation protocol (runner/runner.py:443-444, shapenet_seg.py:327-329): farthest-point sampling 8192 -> 2048 of every generated
    cloud, timed beside the sampling (HIP events, 5 launches after a warm-up) for one cloud and for a batch of 128."""
    from difffacto_amd.pointnet2_ops import pointnet2_utils as pu
    try:
        out = {}
        g = torch.Generator(device=dev).manual_seed(3)
        for B in (1, 128):
            xyz = torch.randn(B, 8192, 3, device=dev, generator=g)
            pu.furthest_point_sample(xyz, 2048)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                idx = pu.furthest_point_sample(xyz, 2048)
            b.record()
            torch.cuda.synchronize()
            out[f"fps_8192_to_2048_B{B}_ms"] = a.elapsed_time(b) / 5
            assert int(idx.max()) < 8192
        return out
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:200]}


def train_iteration(Wnp, B, N, iters=10):
    """Secondary figure (BASELINE configs[4]), outside the timed region of the headline metric: one training
    iteration of the denoiser (forward with saved activations + backward + clip + Adam, bf16 matrix products) on the
    same batch shape, wall-clock over `iters` iterations after three warm-ups."""
    import numpy as np
    import torch
    from difffacto_amd import synth, training
    try:
        dev = "cuda"
        P = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in Wnp.items()}
        pc, mean, logvar, valid = synth.make_latents(B, seed=1)
        seg = synth.make_seg_mask(valid, N)
        var = np.exp(logvar).astype(np.float32)
        idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
        anc, vr = np.take_along_axis(mean, idx, axis=2), np.take_along_axis(var, idx, axis=2)
        rng = np.random.default_rng(0)
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        a = [cu((anc + np.sqrt(vr) * rng.standard_normal((B, 3, N))).astype(np.float32)), cu(rng.integers(0, 1000, size=(B,)).astype(np.int64)),
             cu(pc), cu(np.concatenate([mean, var], 1).astype(np.float32)), cu(anc.transpose(0, 2, 1)), cu(vr.transpose(0, 2, 1)), cu(valid),
             cu(seg.astype(np.int32))]
        noise = cu(rng.standard_normal((B, 3, N)).astype(np.float32))
        opt = training.Adam(list(P.values()), lr=1e-4, max_norm=10.0)

        step = [0]

        last = {}

        def it(p=0.0):
            opt.zero_grad()
            step[0] += 1
            drop = (p, 7000 + step[0]) if p > 0 else None     # a fresh Philox key per step, like the module API draws one (modules.TransformerNet._forward_train)
            loss = training.masked_mse(noise, training.denoiser_train_forward(P, *a, precision="bf16", dropout=drop), None)
            loss.backward()
            last["loss"] = loss.detach()
            opt.step()

        def check_finite(tag):   # after the timed loop (outside it): the last loss, every gradient and every updated parameter
            if not (bool(torch.isfinite(last["loss"])) and all(bool(torch.isfinite(v.grad).all()) and bool(torch.isfinite(v).all()) for v in P.values())):
                raise FloatingPointError(f"train_iteration[{tag}]: non-finite loss / gradient / parameter")
            return float(last["loss"])

        def timed(p):
            for _ in range(3):   # (the block runs behind the chain sweeps: three warm-ups and ten timed iterations keep a 20 ms sample's noise out of the line)
                it(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                it(p)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3

        ms = timed(0.0)
        loss_p0 = check_finite("p=0")
        ms_p02 = timed(0.2)    # configs/train_chair_stage1.py:38 as shipped: nn.Dropout(0.2) behind every to_out and GEGLU, in the fused kernels (k_ff<*, true>)
        out = {"what": "denoiser forward + backward + clip + Adam, bf16 matrix products, fp32 master weights (tools/bench_train.py)",
               "ms": ms, "shapes_per_s": B / ms * 1e3, "batch": B, "npoints": N, "iters": iters, "dropout": 0.0, "finite": True,
               "last_loss": loss_p0, "last_loss_dropout_0.2": check_finite("p=0.2"),
               "dropout_0.2": {"ms": ms_p02, "shapes_per_s": B / ms_p02 * 1e3, "vs_dropout_0": ms_p02 / ms,
                               "what": "the same iteration with the shipped dropout = 0.2 (Philox factors drawn in the fused forward kernel, one bit per element kept for the backward kernels)"}}
        del P, a, noise, opt
        torch.cuda.empty_cache()
        out["stage1"] = stage1_iteration(B, N, iters)
        torch.cuda.empty_cache()
        # BASELINE configs[4] is a bf16 configuration: the same step with bf16 operands for the PointNetV2 trunk's products as well
        # (module.train_precision = "bf16"; the default keeps the encoder in exact fp32)
        out["stage1"]["encoder_bf16_ms"] = stage1_iteration(B, N, iters, encoder_precision="bf16")["ms"]
        torch.cuda.empty_cache()
        s1d = stage1_iteration(B, N, iters, dropout=0.2)                                       # the configuration as shipped
        out["stage1"]["dropout_0.2_ms"], out["stage1"]["dropout_0.2_samples_ms"] = s1d["ms"], s1d["samples_ms"]
        return out
    except Exception as e:   # secondary line: never fail the headline measurement
        return {"error": repr(e)[:200]}


def stage1_iteration(B, N, iters=16, encoder_precision="f32", dropout=0.0):
    """The whole stage-1 training iteration of configs/train_chair_stage1.py (PointNetV2 part encoder in train mode + prior loss
    through the latent flows + denoiser + clip + Adam) through the drop-in modules (examples/train_stage1.py)."""
    import numpy as np
    import torch
    from difffacto_amd import synth, training
    from difffacto_amd.encoders import PartEncoderForTransformerDecoder
    from difffacto_amd.modules import AnchoredDiffusion
    enc = PartEncoderForTransformerDecoder(encoder=dict(type="PointNetV2", zdim=256, per_part_mlp=True), n_class=4, part_aligner=None,
                                           include_z=False, include_part_code=True, include_params=True, use_gt_params=True, kl_weight=5e-4,
                                           use_flow=True, latent_flow_depth=14, latent_flow_hidden_dim=256, gen=True, prior_var=1.0)
    net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=dropout, context_dim=256 + 6,
               n_class=4, class_cond=True, use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True, cat_class_to_x=True)
    diff = AnchoredDiffusion(net=net, num_timesteps=1000, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False,
                             rescale_timesteps=False, model_mean_type="epsilon", learn_variance=True, loss_type='mse', include_anchors=False,
                             precision="bf16")
    enc, diff = enc.cuda().train(), diff.cuda().train()
    enc.encoder.train_precision = encoder_precision
    opt = training.Adam(list(enc.parameters()) + list(diff.model.parameters()), lr=1e-4, max_norm=10.0)
    rng = np.random.Generator(np.random.PCG64(0))
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    _, shift, lv, valid = synth.make_latents(B, seed=0)
    seg = synth.make_seg_mask(valid, N)
    std = np.exp(0.5 * lv).astype(np.float32)
    idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
    pts = (np.take_along_axis(shift, idx, 2) + np.take_along_axis(std, idx, 2) * rng.standard_normal((B, 3, N))).astype(np.float32)
    pcds = {"input": cu(pts.transpose(0, 2, 1)), "ref": cu(pts.transpose(0, 2, 1)), "present": cu(valid), "dp_present": cu(valid),
            "ref_seg_mask": cu(seg.astype(np.int64)), "ref_attn_map": cu(np.eye(4, dtype=np.float32)[seg]), "part_shift": cu(shift),
            "part_scale": cu(std), "noise": torch.zeros(B, 32).cuda()}

    def it():
        opt.zero_grad()
        losses = training.stage1_losses(enc, diff, pcds)
        sum(v.sum() for k, v in losses.items() if "loss" in k).backward()
        opt.step()

    for _ in range(3):   # (short samples of this loop scatter by +-5 %: 12.2 ms over 24 iterations read 12.5 .. 13.8 over 4 .. 8)
        it()
    samples = []
    for _ in range(3):   # three samples of `iters` iterations, the fastest one is the figure: ~470 launches per iteration, 6.7 ms of host enqueue time against
        torch.cuda.synchronize()   # ~10 ms of GPU time (tools/experiments/stage1_host_time.py); samples differ by +-0.7 ms, one in ten reads 30-40 % high (r05: 15.1 vs 11.1 ms)
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        samples.append((time.perf_counter() - t0) / iters * 1e3)
    ms = min(samples)
    return {"what": "PointNetV2 (train mode, fp32) + prior loss through 4 x 14 coupling layers + denoiser (bf16 products) + clip + Adam",
            "ms": ms, "shapes_per_s": B / ms * 1e3, "samples_ms": samples}


def measured_traffic(T, B, N):
    """Fabric-side bytes per launch from the newest committed PMC profile (profiles/rNN_traffic.json: separate rocprofv3 --pmc
    passes of this command, see profiles/README.md — counters cannot be read from inside the run).  Round 5 measured at TWO chain lengths
    (T = 20 and 40, tools/prof_write_size.sh): traffic(T) = fixed + per_step * T — the writes and most of the reads are a per-launch constant
    (WRITE_SIZE does not grow with T at all, profiles/r05_write_size_scaling.txt); older files hold one T = 20 launch divided by 20.
    Returns (bytes or None, provenance or None): the bench line names the file and round it came from."""
    best = _newest_profile("traffic.json")
    if best is None:
        return None, None
    try:
        d = json.load(open(best[1]))
        if d["B"] == B and d["N"] == N:
            fixed = d.get("bytes_per_launch_fixed", 0.0)
            return fixed + d["bytes_per_step"] * T, {"file": os.path.relpath(best[1], ROOT), "round": d.get("round", f"r{best[0]:02d}"),
                                                     "T_profiled": d.get("T_profiled"), "bytes_per_launch_fixed": fixed, "bytes_per_step": d["bytes_per_step"],
                                                     "note": "PMC passes of an earlier run of this command at two chain lengths, fixed + per_step * T; not measured in this run"}
    except Exception:
        pass
    return None, None


def cpu_baseline(W, N, budget_s=20.0):
    """PyTorch-CPU restatement of the reference's p_sample (oracle/torch_cpu.py: the same unfused op sequence the
    reference's modules run, pinned to the reference goldens in tests/test_oracle_golden.py) timed on this box's host cores
    with torch's own intra-op threading, on a bounded sample: B shapes x a few steps of the same workload; the chain is
    strictly sequential in t, so shapes/s at T=1000 is extrapolated linearly from s/step/shape."""
    from oracle import diffusion as odf
    from oracle import torch_cpu as tc
    B, Tc = 16, 2
    part_code, mean, logvar, valid = synth.make_latents(B, seed=7)
    var = np.exp(logvar).astype(np.float32)
    seg = synth.make_seg_mask(valid, N)
    anchors, variance = odf.gather_params(seg, mean, var)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ctx = [tt(part_code), tt(np.concatenate([mean, var], 1))]
    Wt = {k: tt(v) for k, v in W.items()}
    tb = odf.Tables(1000)
    g = torch.Generator().manual_seed(0)
    anchors, variance, seg_t, valid_t = tt(anchors), tt(variance), tt(seg), tt(valid)
    x = torch.sqrt(variance) * torch.randn(B, 3, N, generator=g) + anchors
    z = torch.randn(B, 3, N, generator=g)
    def steps(k, x):
        t0 = time.perf_counter()
        for i in range(k):
            x, _ = tc.p_sample(tb, Wt, x, 999 - i, anchors, ctx, variance, seg_t, valid_t, z)
        return time.perf_counter() - t0, x

    default_threads = torch.get_num_threads()
    with torch.no_grad():
        # torch's default (one thread per physical core) oversubscribes these mid-size GEMMs on a 128-core host: probe a few
        # thread counts with one step each and keep the fastest
        best = None
        for nt in sorted({min(default_threads, c) for c in (8, 16, 32, 64, 128)}):
            torch.set_num_threads(nt)
            steps(1, x)                          # warm-up (thread pool, page-in)
            d1, _ = steps(1, x)
            if best is None or d1 < best[0]:
                best = (d1, nt)
        threads = best[1]
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        n = 0
        while True:
            d, x = steps(Tc, x)
            n += Tc
            if time.perf_counter() - t0 > budget_s / 2 or n >= 40:
                break
    torch.set_num_threads(default_threads)
    dt = time.perf_counter() - t0
    s_per_step_shape = dt / (n * B)
    return {"value": 1.0 / (s_per_step_shape * 1000), "unit": "shapes/s", "cores": threads, "kind": "port",
            "sample": f"PyTorch-CPU restatement of the reference's p_sample (oracle/torch_cpu.py), fp32, {threads} intra-op threads (fastest of the probed counts) of "
                      f"{os.cpu_count()} logical cores, B={B} shapes x {n} steps at N={N} ({dt:.1f}s), "
                      f"{s_per_step_shape * 1e3:.1f} ms/step/shape extrapolated to T=1000",
            "ms_per_step_per_shape": s_per_step_shape * 1e3}


def parity_block(Wnp, N, T=100, B=2, B_gpu=32):
    """BASELINE.md §3 / SURVEY.md §8(d) 'quality parity', from the same run and outside the timed region: the bf16 HIP chain
    and the exact-fp32 HIP chain against the PyTorch-CPU oracle (oracle/torch_cpu.py, pinned to the reference goldens) on
    IDENTICAL explicit noise, T steps: max-abs point deviation, Chamfer-L2 and auction EMD (the evaluation's setting 0.002 / 10000
    on unit-box-normalised clouds; libdfx's own metric kernels).  The HIP side runs B_gpu = 32 shapes x N points — a batch at which
    the launcher takes the HEADLINE kernels (k_denoise_pipe<8> / k_denoise_pipe_f32<8>, one workgroup per CU; the variant that ran is
    reported per precision) — and the oracle walks the first B of them (shapes are independent)."""
    from oracle import diffusion as odf
    from oracle import torch_cpu as tc
    from difffacto_amd.engine import DenoiserEngine, last_kernel_variant
    from difffacto_amd.metrics import EMD, chamfer_l2
    try:
        pc_g, mean_g, logvar_g, valid_g = synth.make_latents(B_gpu, seed=5)
        var_g = np.exp(logvar_g).astype(np.float32)
        seg_g = synth.make_seg_mask(valid_g, N)
        g = torch.Generator().manual_seed(T)
        xT_g = torch.randn(B_gpu, 3, N, generator=g)
        zs_g = torch.randn(T, B_gpu, 3, N, generator=g)
        pc, mean, var, valid, seg = pc_g[:B], mean_g[:B], var_g[:B], valid_g[:B], seg_g[:B]
        xT, zs = xT_g[:B], zs_g[:, :B]
        anchors, variance = odf.gather_params(seg, mean, var)
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        Wt = {k: tt(v) for k, v in Wnp.items()}
        tb = odf.Tables(T)
        ctx = [tt(pc), tt(np.concatenate([mean, var], 1))]
        t0 = time.perf_counter()
        with torch.no_grad():
            x = torch.sqrt(tt(variance)) * xT + tt(anchors)
            for i, t in enumerate(range(T - 1, -1, -1)):
                x, _ = tc.p_sample(tb, Wt, x, t, tt(anchors), ctx, tt(variance), tt(seg), tt(valid), zs[i])
        ref = x.transpose(1, 2).contiguous().cuda()
        cpu_s = time.perf_counter() - t0
        out = {"what": f"HIP chain ({B_gpu} shapes x {N} pts in the launch) vs PyTorch-CPU oracle on identical explicit noise, compared on the "
                       f"first {B} shapes, T={T}; part sigma ~{float(np.sqrt(var).mean()):.3f}; EMD on unit-box-normalised clouds (eps 0.002, "
                       f"10000 iterations)",
               "oracle_cpu_s": cpu_s}

        def emd_pair(x, y):
            lo = torch.minimum(x.amin((1, 2), keepdim=True), y.amin((1, 2), keepdim=True))
            hi = torch.maximum(x.amax((1, 2), keepdim=True), y.amax((1, 2), keepdim=True))
            return EMD(0.002, 10000, True)(((x - lo) / (hi - lo)).contiguous(), ((y - lo) / (hi - lo)).contiguous())
        for prec in ("bf16", "f32"):
            eng = DenoiserEngine(Wt, T, precision=prec)
            c = eng.prepare_shapes(tt(pc_g), tt(mean_g), tt(var_g), tt(valid_g))
            pred, _ = eng.sample_chain(c, tt(seg_g), x_T_noise=xT_g, step_noise=zs_g)
            variant = last_kernel_variant()
            eng.close()
            pred = pred[:B].contiguous()
            out[prec] = {"kernel_variant": variant, "max_abs": float((pred - ref).abs().max()), "mean_abs": float((pred - ref).abs().mean()),
                         "chamfer_l2": float(chamfer_l2(pred, ref).mean()), "emd": float(emd_pair(pred, ref).mean())}
        return out
    except Exception as e:   # secondary block: never fail the headline measurement
        return {"error": repr(e)[:200]}


def chain_line(params, names, B, N, T, precision, dev, noise_scale=100.0, launches=1, seed=99):
    """One configuration of the same hot path, outside the timed region of the headline: latents (the config's noise_scale) +
    context preparation + ONE chain launch.  Reports the HIP-event time of the chain launch (kernel figure, roofline fraction)
    and the wall clock of the whole pass (latents and preparation included) after a warm-up evaluation of the same kernel."""
    from difffacto_amd.engine import DenoiserEngine
    from difffacto_amd.latents import LatentSampler
    try:
        eng = DenoiserEngine({k: params[k] for k in names}, num_timesteps=T, precision=precision, device=dev)
        sampler = LatentSampler({k[len("encoder."):]: v for k, v in params.items() if k.startswith("encoder.")},
                                noise_scale=noise_scale, device=dev)
        torch.cuda.manual_seed(seed)
        valid = torch.from_numpy(synth.make_latents(B, seed=1000 + seed)[3].copy()).to(dev)

        # the pass as a service runs it (difffacto_amd/pipeline.py): front end = draws + latent sampler + context preparation as ONE
        # hipGraph, replayed on a side stream beside the previous batch's chain; the chain launch waits on an event, not on the host
        from difffacto_amd.pipeline import SamplingPipeline
        pipe = SamplingPipeline(eng, sampler, B, N, valid)
        # warm-up (code load, LDS attribute, allocator): a whole pass when that is cheap, otherwise one evaluation through the same
        # kernel instantiation
        if T <= 100:
            for _ in pipe.run(1, seed0=1):
                pass
        else:
            lat = pipe.latents(0)
            eng.eps(pipe.inst[0].ctx, torch.zeros(B, 3, N, device=dev), lat["seg_mask"], T - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        preds = list(pipe.run(launches, seed0=2, time_chain=True))
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / launches * 1e3
        ok = all(bool(torch.isfinite(p).all()) for p in preds)   # every timed launch
        ms = float(np.mean([a.elapsed_time(b) for a, b in pipe.last_chain_events]))
        from difffacto_amd.engine import last_kernel_variant
        variant, fold = last_kernel_variant(), (eng.w1_fold() + (eng.w1_fold_channel(),) if precision == "bf16" else None)
        eng.close()
        ach = flops_per_step(N) * T * B / (ms * 1e-3) / 1e12
        return {"batch": B, "npoints": N, "num_timesteps": T, "dtype": precision, "noise_scale": noise_scale, "kernel_ms": ms,
                "shapes_per_s": B / ms * 1e3, "wall_ms": wall_ms, "wall_shapes_per_s": B / wall_ms * 1e3, "achieved_tflops": ach,
                "frac": ach / PEAK_TFLOPS[precision], "launches": launches, "finite": ok, "kernel_variant": variant,
                **({} if fold is None else {"w1_fold": {"folded": fold[0], "ratio": round(fold[1], 3), "channel": fold[2]}})}
    except Exception as e:
        return {"error": repr(e)[:200]}


def t100_line(params, names, B, N, precision, dev):
    """The shipped configs run num_timesteps=100 (configs/gen_chair.py:88): the same pass at T=100 — kernel time of the chain launch
    and the wall figure with the latent sampler and the context preparation included."""
    return chain_line(params, names, B, N, 100, precision, dev, launches=5)


def f32_line(params, names, B, N, T, dev):
    """Reference precision (the reference computes in fp32 end to end, attention.py:296-306): the same chain on the exact-fp32
    persistent kernel k_denoise_pipe_f32 (v_mfma_f32_32x32x2_f32, bit-identical to the direct fp32 kernel of the parity gates)."""
    r = chain_line(params, names, B, N, T, "f32", dev)
    r["kernel"] = "k_denoise_pipe_f32"
    return r


def sweep_block(params, names, precision, dev, T):
    """BASELINE configs[2] (gen_airplane / gen_car / gen_lamp: the configs differ from gen_chair in the aligner's noise_scale and, for
    gen_car, npoints = 8192; configs/gen_*.py:29,88-90) and configs[1]'s batch sweep, one chain launch each at the headline T."""
    out = {}
    for name, ns, n, b in (("gen_airplane", 50.0, 2048, 128), ("gen_car", 50.0, 8192, 128), ("gen_lamp", 10.0, 2048, 128),
                           ("gen_chair_B1024", 100.0, 2048, 1024)):
        out[name] = chain_line(params, names, b, n, T, precision, dev, noise_scale=ns)
    # a checkpoint-like hazard: LayerNorm gain outlier in the channel the W1 bias fold used to sit on — round 5 sent such an
    # engine to the ~3x slower direct kernel; since round 6 dfx_denoiser_create moves the fold to another hidden channel and the headline kernel stays
    po = dict(params)
    g3 = params["transformer_blocks.2.norm3.weight"].clone()
    g3[127] *= 64.0
    po["transformer_blocks.2.norm3.weight"] = g3
    out["outlier_weights"] = chain_line(po, names, 128, 2048, T, precision, dev)
    if isinstance(out["outlier_weights"], dict) and isinstance(out.get("gen_lamp"), dict) and "kernel_ms" in out["outlier_weights"] and "kernel_ms" in out["gen_lamp"]:
        out["outlier_weights"]["what"] = "gen_chair B = 128 with transformer_blocks.2.norm3.weight[127] x 64"
        out["outlier_weights"]["vs_ordinary_weights"] = out["outlier_weights"]["kernel_ms"] / out["gen_lamp"]["kernel_ms"]   # (same B, N, T, kernel)
    if isinstance(out.get("gen_car"), dict):
        out["gen_car"].update(fps_line(dev))   # the evaluation protocol's FPS 8192 -> 2048 leg, next to the sampling it follows
    return out


def small_batch_line(params, names, sampler, N, precision, dev, T):
    """Latency side of the same kernel family (SURVEY.md §8(d) config 2, B = 1): one T-step chain for ONE shape, for four and for eight, HIP-event
    time of the chain launch; the co-operative kernels (one tile per workgroup up to B = 4, two tiles for B = 5 .. 8) take these sizes,
    bit-identical to the pipelined one (tested); `kernel_variant` = what the launcher chose."""
    from difffacto_amd.engine import DenoiserEngine
    try:
        out = {"num_timesteps": T}
        eng = DenoiserEngine({k: params[k] for k in names}, num_timesteps=T, precision=precision, device=dev)
        from difffacto_amd.pipeline import SamplingPipeline
        from difffacto_amd.engine import last_kernel_variant
        for B in (1, 4, 8):
            torch.cuda.manual_seed(98)
            pipe = SamplingPipeline(eng, sampler, B, N, torch.ones(B, 4, device=dev))
            for _ in pipe.run(1, seed0=1):     # warm-up pass
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe.run(1, seed0=2, time_chain=True):   # ONE request: front end (one graph launch) -> chain, nothing to overlap with
                pass
            torch.cuda.synchronize()
            wall_one = (time.perf_counter() - t0) * 1e3
            a, b = pipe.last_chain_events[0]
            ms = a.elapsed_time(b)
            t0 = time.perf_counter()
            for _ in pipe.run(3, seed0=3):     # a stream of requests: the next front end rides beside the running chain
                pass
            torch.cuda.synchronize()
            wall_stream = (time.perf_counter() - t0) / 3 * 1e3
            out[f"B{B}"] = {"ms_per_chain": ms, "shapes_per_s": B / ms * 1e3, "wall_ms_one_pass": wall_one, "wall_ms_per_pass_streamed": wall_stream,
                            "kernel_variant": last_kernel_variant()}
        eng.close()
        return out
    except Exception as e:
        return {"error": repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="shapes per GPU (shipped val batch_size=128)")
    ap.add_argument("--total-shapes", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]: 1024 shapes over the node): the JOB is this many shapes, block-partitioned "
                         "over the ranks (overrides --batch; scaling = 'strong')")
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-line", action="store_true", help="skip the secondary training-iteration measurement")
    ap.add_argument("--force-direct", action="store_true")
    ap.add_argument("--pipe-waves", type=int, default=0, help="A/B: force a chain-kernel variant (dfx_debug_pipe_waves: 8 / 4 / 2 / 1 / 64)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the secondary blocks: parity (HIP vs PyTorch-CPU oracle, T=100, 2 shapes), t100, f32, sweep, small_batch")
    ap.add_argument("--dump-clouds", default=None, help="rank 0 writes the last step's gathered clouds (shapes, N, 3) to this .npy")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus > 1:
        import torch.distributed as dist
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        backend = os.environ.get("DFX_BENCH_BACKEND", "nccl")   # "gloo": plumbing test of the N>1 path on one GPU
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % ndev)
        if backend == "nccl":
            # day-one evidence of what RCCL did: its version banner and the transport of every channel ("via P2P/IPC" = xGMI peer
            # access, "via SHM" = host bounce) go to a per-process file that rank 0 summarises into the JSON line
            if "NCCL_DEBUG" not in os.environ and "NCCL_DEBUG_FILE" not in os.environ:
                os.environ["NCCL_DEBUG"] = "INFO"
                os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,P2P,SHM,NET"
                os.environ["NCCL_DEBUG_FILE"] = f"/tmp/dfx_rccl_{os.getpid()}.log"
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from difffacto_amd.engine import DenoiserEngine
    from difffacto_amd.parallel import broadcast_params, collective_library, gather_clouds, probe_gather

    N, T = args.npoints, args.timesteps
    # The JOB: `total` shapes, block-partitioned over the ranks (parallel.shard_range).  Weak scaling (default): --batch shapes per
    # GPU, total = world * batch.  Strong scaling (--total-shapes S): total = S whatever the world size, rank r takes its block.
    from difffacto_amd.parallel import shard_range
    total = args.total_shapes if args.total_shapes > 0 else args.batch * world
    lo, hi = shard_range(total, rank, world)
    B = hi - lo
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    assert B >= 1, f"rank {rank} has no shape: --total-shapes {total} < world size {world}"
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_force_direct(int(args.force_direct))
    _ffi.lib().dfx_debug_pipe_waves(args.pipe_waves)
    names = [n for n, _ in synth.denoiser_param_shapes()]
    lat_shapes = [("encoder." + n, s) for n, s in synth.latent_param_shapes()]
    if rank == 0:
        Wnp = synth.make_denoiser_weights(seed=0)
        Lnp = synth.make_latent_weights(seed=0)
        params = {k: torch.from_numpy(Wnp[k]).to(dev) for k in names}
        params.update({"encoder." + k: torch.from_numpy(v).to(dev) for k, v in Lnp.items()})
    else:
        Wnp = None
        params = {k: torch.empty(s, dtype=torch.float32, device=dev)
                  for k, s in list(synth.denoiser_param_shapes()) + lat_shapes}
    bcast_ms, bcast_bytes, world_seen = 0.0, 0, None
    if dist is not None:
        from difffacto_amd.parallel import describe_world
        world_seen = describe_world(dev)          # backend, world size and device of every rank as the process group reports them
        assert world_seen["world_size"] == world
        gather_info = probe_gather(dev)           # one tiny rooted gather; all ranks fall back to all_gather together if it raises
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        broadcast_params(params, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
        bcast_bytes = 4 * sum(int(v.numel()) for v in params.values())
    eng = DenoiserEngine({k: params[k] for k in names}, num_timesteps=T, precision=args.precision, device=dev)
    from difffacto_amd.latents import LatentSampler
    sampler = LatentSampler({k[len("encoder."):]: v for k, v in params.items() if k.startswith("encoder.")},
                            noise_scale=100.0, device=dev)          # configs/gen_chair.py:14-31

    # Synthetic part-presence patterns, resident in HBM.  Rank r owns global shapes [lo, hi): every per-shape input — presence
    # pattern, latent draws, the chain's Philox stream (keyed by the global point id through shape_offset) — is a function of the
    # GLOBAL shape index only, so the generated clouds do not depend on the number of GPUs.
    valid = torch.from_numpy(synth.make_latents(total, seed=1000)[3][lo:hi].copy()).to(dev)
    gen = torch.Generator(device=dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def one_step(i, timed):
        gen.manual_seed(1234 + i)                                        # the whole job's draws, then this rank's block
        w = torch.randn(total, 256, 4, device=dev, generator=gen)[lo:hi]    # part_encoders.py:1054
        an = torch.randn(total, 32, device=dev, generator=gen)[lo:hi]       # :1065 (K = 1 noise per shape)
        lat = sampler.sample_latents(w, an, valid, K=1, npoints=N)
        ctx = eng.prepare_shapes(lat["part_code"], lat["params"][:, :3], lat["params"][:, 3:], lat["valid_id"])
        if timed:
            ev[i][0].record()
        pred, _ = eng.sample_chain(ctx, lat["seg_mask"], seed=7000 + i, shape_offset=lo)
        if timed:
            ev[i][1].record()
        if dist is not None:
            pred = gather_clouds(pred, dst=0, sizes=sizes)   # block partition: sizes known, no size exchange / host sync
        return pred

    for i in range(args.warmup):
        one_step(i, False)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    box = BoxSampler(torch.cuda.current_device()).start() if rank == 0 else None
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = one_step(i, True)
    fence()
    t1 = time.perf_counter()
    dt = t1 - t0
    calib = None
    if rank == 0:
        box.stop()
        if world == 1 and args.precision == "bf16":
            calib = box_calibration()      # right behind the timed region: same box, same thermal state
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    from difffacto_amd.engine import last_kernel_variant
    variant = last_kernel_variant()           # the kernel the timed launches took (dfx_last_kernel_variant)
    rank_ms = None
    if dist is not None:
        from difffacto_amd.parallel import all_gather_floats
        rank_ms = all_gather_floats([kern_ms, dt * 1e3 / args.steps], dev)   # per rank: chain kernel ms, wall ms per step

    if rank == 0:
        assert out is not None and torch.isfinite(out).all()
        assert out.shape[0] == total, (out.shape, total)
        if args.dump_clouds:
            np.save(args.dump_clouds, out.cpu().numpy())
        value = total * args.steps / dt
        F = flops_per_step(N) * T * B                      # algorithmic FLOPs per launch (one rank)
        traffic, traffic_src = measured_traffic(T, B, N) if args.precision == "bf16" else (None, None)
        achieved = F / (kern_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        res = {
            "metric": "generated shapes/sec (2048 pts, 1000-step DDPM)",
            "value": value, "unit": "shapes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.total_shapes > 0 else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"gen_chair decode: {B} shapes/GPU x {N} pts x 4 parts, T={T} DDPM steps, "
                                   f"random-init flows + part aligner + denoiser (depth 5, inner 128), in-kernel Philox noise",
                       "batch_per_gpu": B, "npoints": N, "num_timesteps": T, "parallelism": f"dp{world} (independent shapes)",
                       "total_shapes": total, "weights_bcast_ms": bcast_ms, "weights_bcast_bytes": bcast_bytes},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": variant + " (persistent T-step chain)", "kernel_variant": variant, "kernel_ms": kern_ms,
                         "flops_per_launch": F},
        }
        res["roofline"]["box"] = box.summary(t0, t1)   # clock / power / temperature samples taken DURING the timed region
        if args.precision == "bf16":
            res["roofline"]["power_limited_ceiling"] = power_limited_ceiling()
            if calib is not None and "bare_mfma_tflops" in calib:
                # executed MFMA flops of the chain kernel: 2000 MFMAs per wavefront-step of 32 points (counted: SQ_INSTS_MFMA, profiles/r04_pmc_chain_T20_B128.txt)
                executed = 2000.0 * 32768.0 * (B * N / 32.0) * T / (kern_ms * 1e-3) / 1e12
                res["roofline"]["box_bare_mfma_tflops"] = calib["bare_mfma_tflops"]
                res["roofline"]["executed_tflops"] = executed
                res["roofline"]["frac_of_box"] = executed / calib["bare_mfma_tflops"]
                res["roofline"]["frac_of_box_algorithmic"] = achieved / calib["bare_mfma_tflops"]
            res["roofline"]["box_calibration"] = calib
        res["config"]["shapes_gathered"] = int(out.shape[0])
        if world_seen is not None:   # what the process group itself reports: the SCALE record shows the collective library saw N ranks
            res["config"].update({"backend": world_seen["backend"], "world_size_seen": world_seen["world_size"],
                                  "rank_devices": world_seen["devices"], "visible_gpus": torch.cuda.device_count(),
                                  "gather": gather_info["gather"], "gather_fallback": gather_info["fallback"], "shapes_per_rank": sizes,
                                  "collective_library": collective_library(), "transport": rccl_transport_summary(),
                                  "rank_kernel_ms": [r[0] for r in rank_ms], "rank_wall_ms_per_step": [r[1] for r in rank_ms]})
        if world == 1 and not args.no_parity:
            res["parity"] = parity_block(Wnp, N)
            res["t100"] = t100_line(params, names, B, N, args.precision, dev)
            if args.precision == "bf16":
                res["f32"] = f32_line(params, names, B, N, T, dev)
            res["sweep"] = sweep_block(params, names, args.precision, dev, T)
            res["small_batch"] = small_batch_line(params, names, sampler, N, args.precision, dev, T)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(Wnp, N)
        if world == 1 and not args.no_train_line and args.precision == "bf16":
            res["train_iteration"] = train_iteration(Wnp, B, N)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

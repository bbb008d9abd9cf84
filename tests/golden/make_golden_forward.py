#!/usr/bin/env python
"""Golden vectors of the reference's top-level compositions (dev container only; same conventions as make_golden.py).

    python tests/golden/make_golden_forward.py

forward_gen_B2_K2_T10.npz     AnchorDiffAE.forward, eval + gen branch (anchor_gen.py:970-1084): encoder pass, sample_latents, decode,
                              the K-fold "{k}_sample {i}" / "sample prior {i}" dict of the cIMLE configuration (configs/gen_chair.py with
                              num_timesteps = 10, npoints = 64, cimle_sample_num = 2, ret_interval = 5)
forward_sample_B2_K2_T10.npz  the same class with gen = False: the encode -> decode "sample" mode (:1085-1134: sample_noise, encoder forward
                              with a (B, K, 32) noise, decode, per-sample dict)
encoder_fwd_B3_N96.npz        PartEncoder.forward with the part aligner (part_encoders.py:1185-1260; fit_loss_type 4, kl_weight 5e-4 so
                              that the prior loss is exercised) and sample_noise (:388-414)
train_loop_B3_N64_T10.npz     three iterations of the reference's training loop on the denoiser (runner/runner.py:299-316: zero_grad,
                              training_losses, backward, clip_grad_norm_(10), Adam.step, LinearLR.step), dropout 0, fp32: per-iteration
                              loss, gradient norm, learning rate, parameter checksums and 256 sampled elements per parameter
                              (values after the step + the clipped gradient the step consumed)

aligner_grads_B5.npz          torch autograd through the reference's PartAlignerTransformer (outputs, d / d parameters, d / d part_code)
stage2_step_B4_N64_T10.npz    one stage-2 training forward + backward of the reference's AnchorDiffAE in train() mode (part aligner + fit loss + masked
                              MSE, Dropout 0): loss dict, total, d total / d (aligner parameters), the BatchNorm running statistics left behind

Every torch.randn / torch.randn_like the reference executes is served from a numpy PCG64 stream and RECORDED in call order
("draw_{i}"); the tests replay the draws at the same sites of the mirror (the T + 1 chain draws are handed to decode as explicit
x_T / step noise).  Weights are not stored (difffacto_amd.synth regenerates them from seeds on both sides).
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from difffacto_amd import synth  # noqa: E402
from make_golden import load_denoiser_weights, load_latent_weights, make_case, to_ref_inputs  # noqa: E402

F32 = np.float32


class DrawRecorder:
    """Serves torch.randn / randn_like from one numpy stream and records every draw in call order."""

    def __init__(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.draws = []

    def _draw(self, shape):
        a = self.rng.standard_normal(tuple(int(s) for s in shape)).astype(F32)
        self.draws.append(a)
        return torch.from_numpy(a.copy())

    def __enter__(self):
        self._real = (torch.randn, torch.randn_like)

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            return self._draw(shape)

        torch.randn, torch.randn_like = randn, lambda x, **kw: self._draw(x.shape)
        self._cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t          # hard-coded .cuda() in the reference (part_encoders.py:1137,1176): no CUDA here
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._real
        torch.Tensor.cuda = self._cuda

    def as_dict(self):
        return {f"draw_{i}": a for i, a in enumerate(self.draws)}


def load_all_weights(model):
    load_denoiser_weights(model, synth.make_denoiser_weights(seed=0))
    load_latent_weights(model, synth.make_latent_weights(seed=0))
    enc = model.encoder.encoder
    W = synth.make_pointnet_v2_weights(seed=0)
    sd = enc.state_dict()
    for k, a in W.items():
        assert tuple(sd[k].shape) == a.shape, k
        sd[k] = torch.from_numpy(a.copy())
    enc.load_state_dict(sd)


def make_batch(B, N, seed, absent=((1, 3),)):
    """A synthetic val batch with the keys AnchorDiffAE.forward / PartEncoder.forward read (shapenet_seg.py:430-520 names)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    present = np.ones((B, 4), F32)
    for b, j in absent:
        present[b, j] = 0
    seg = np.zeros((B, N), np.int64)
    for b in range(B):
        ids = np.flatnonzero(present[b])
        seg[b] = ids[rng.integers(0, len(ids), size=N)]
        seg[b, :len(ids)] = ids                                   # every present part owns at least one point
    attn = np.eye(4, dtype=F32)[seg]
    part_shift = (rng.standard_normal((B, 3, 4)) * 0.3).astype(F32)
    part_scale = rng.uniform(0.2, 0.6, size=(B, 3, 4)).astype(F32)
    idx = np.broadcast_to(seg[:, None, :], (B, 3, N))
    ref = (np.take_along_axis(part_shift, idx, 2) + np.take_along_axis(part_scale, idx, 2) * rng.standard_normal((B, 3, N))).astype(F32)
    ref = np.ascontiguousarray(ref.transpose(0, 2, 1))
    batch = dict(input=ref.copy(), ref=ref, seg_mask=seg.copy(), ref_seg_mask=seg, attn_map=attn.copy(), ref_attn_map=attn, present=present,
                 noise=rng.standard_normal((B, 32)).astype(F32), shift=(rng.standard_normal((B, 1, 3)) * 0.1).astype(F32),
                 scale=rng.uniform(0.8, 1.2, size=(B, 1, 1)).astype(F32), part_shift=part_shift, part_scale=part_scale,
                 token=rng.integers(0, 100, size=(B, 8)).astype(np.int64))
    return batch


def to_torch(batch):
    return {k: torch.from_numpy(v.copy()) for k, v in batch.items()}


def np_out(d):
    out = {}
    for k, v in d.items():
        v = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        out[f"out/{k}"] = v.astype(F32) if v.dtype.kind == "f" else v
    return out


def gen_forward(tag, gen, B=2, K=2, N=64, T=10, seed=101):
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = ref_import.build_reference_model("gen_chair.py", num_timesteps=T)
    load_all_weights(model)
    model.eval()
    model.npoints, model.cimle_sample_num, model.ret_traj, model.ret_interval, model.gen = N, K, True, 5, gen
    batch = make_batch(B, N, seed)
    with DrawRecorder(seed + 1) as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out = model(to_torch(batch), device="cpu", epoch=0)
    assert len(out) == 1
    pred, name = out[0]
    np.savez_compressed(os.path.join(HERE, f"forward_{tag}.npz"), **{f"in/{k}": v for k, v in batch.items()}, **rec.as_dict(), **np_out(pred),
                        name=np.array(name), K=np.array(K), T=np.array(T), ret_interval=np.array(5), n_draws=np.array(len(rec.draws)))
    print(f"wrote forward_{tag}: '{name}',", len(pred), "keys,", len(rec.draws), "draws, |pred| max", float(pred["pred"].abs().max()))
    print("   keys:", sorted(pred, key=str))
    print("   draw shapes:", [a.shape for a in rec.draws[:4]], "...", rec.draws[-1].shape)


def gen_encoder_forward(tag, B=3, N=96, seed=111, num=3):
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = ref_import.build_reference_model("gen_chair.py", num_timesteps=10)
    load_all_weights(model)
    model.eval()
    enc = model.encoder
    enc.kl_weight = 5e-4
    batch = make_batch(B, N, seed, absent=((0, 2), (2, 0)))
    out = {}
    with DrawRecorder(seed + 1) as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ctx, mpp, lpp, fpp, losses, lat = enc(to_torch(batch), "cpu", epoch=7)
        noise, idx = enc.sample_noise(to_torch(batch), "cpu", num)
        ctx2, mpp2, lpp2, fpp2, losses2, lat2 = enc(to_torch(batch), "cpu", noise=noise, epoch=7)
    out.update({"ctx0": ctx[0], "ctx1": ctx[1], "mean_pp": mpp, "logvar_pp": lpp, "flag_pp": fpp, "part_code": lat[0], "mean": lat[1],
                "logvar": lat[2], "noise": lat[3], "sn_noise": noise, "sn_id": idx, "k_ctx0": ctx2[0], "k_ctx1": ctx2[1], "k_mean_pp": mpp2,
                "k_logvar_pp": lpp2, "k_flag_pp": fpp2, "k_fit_loss": losses2["fit_loss"]})
    out.update({"loss/" + k: v for k, v in losses.items()})
    np.savez_compressed(os.path.join(HERE, f"encoder_fwd_{tag}.npz"), **{f"in/{k}": v for k, v in batch.items()}, **rec.as_dict(), **np_out(out),
                        num=np.array(num), epoch=np.array(7), kl_weight=np.array(5e-4), n_draws=np.array(len(rec.draws)))
    print(f"wrote encoder_fwd_{tag}:", {k: float(v.float().abs().max()) for k, v in losses.items()}, "| part_code max", float(lat[0].abs().max()),
          "| sample_noise ids", idx.tolist())


def gen_train_loop(tag, B=3, N=64, T=10, seed=121, iters=3):
    """runner/runner.py:299-316 on the denoiser alone (the encoder's outputs are data here): Adam(lr 2e-3), clip 10, LinearLR stepped per
    iteration (the runner's `max_epoch is None` mode) from 2e-3 to 1e-4 over epochs 0..2."""
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = ref_import.build_reference_model("gen_chair.py", num_timesteps=T)
    load_denoiser_weights(model, synth.make_denoiser_weights(seed=0))
    from difffacto.optimizers.schedulers import LinearLR
    net = model.diffusion.model
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sched_cfg = dict(start_epoch=0, end_epoch=2, start_lr=2e-3, end_lr=1e-4)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3, weight_decay=0.)
    sched = LinearLR(opt, **sched_cfg)
    case = make_case(B, N, seed, False)
    anchors, variance, ctx, va, sg = to_ref_inputs(*case)
    rng = np.random.Generator(np.random.PCG64(seed + 300))
    srng = np.random.Generator(np.random.PCG64(4243))
    names = [n for n, _ in net.named_parameters()]
    sample_idx = {n: np.sort(srng.choice(p.numel(), size=min(256, p.numel()), replace=False)).astype(np.int64) for n, p in net.named_parameters()}
    out = {"x_start": [], "noise": [], "t": [], "loss": [], "grad_norm": [], "lr": [], "param_sum": [], "param_l2": []}
    samples = {n: [] for n in names}
    gsamples = {n: [] for n in names}
    for it in range(iters):
        x0 = (np.sqrt(variance.numpy()) * rng.standard_normal((B, 3, N)).astype(F32) * 0.5 + anchors.numpy()).astype(F32)
        noise = rng.standard_normal((B, 3, N)).astype(F32)
        t = rng.integers(0, T, size=(B,)).astype(np.int64)
        opt.zero_grad()
        r = model.diffusion.training_losses(torch.from_numpy(x0), torch.from_numpy(t), anchors=anchors, variance=variance, ctx=ctx,
                                            anchor_assignment=sg.to(torch.int32), valid_id=va, flags=None, noise=torch.from_numpy(noise))
        loss = r["mse_loss"]
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 10)
        for n, p in net.named_parameters():     # the (clipped) gradient Adam consumes, at the sampled positions
            gsamples[n].append(p.grad.reshape(-1)[torch.from_numpy(sample_idx[n])].numpy().astype(F32))
        out["lr"].append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        out["x_start"].append(x0), out["noise"].append(noise), out["t"].append(t)
        out["loss"].append(float(loss)), out["grad_norm"].append(float(gn))
        with torch.no_grad():
            flat = torch.cat([p.reshape(-1).double() for p in net.parameters()])
            out["param_sum"].append(float(flat.sum())), out["param_l2"].append(float(flat.norm()))
            for n, p in net.named_parameters():
                samples[n].append(p.detach().reshape(-1)[torch.from_numpy(sample_idx[n])].numpy().astype(F32))
    model.eval()
    arrays = {k: np.stack(v) if k in ("x_start", "noise", "t") else np.array(v, np.float64) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, f"train_loop_{tag}.npz"), part_code=case[0], mean=case[1], logvar=case[2], valid=case[3], seg=case[4],
                        weight_seed=np.array(0), iters=np.array(iters), max_norm=np.array(10.0), sched=np.array([0, 2, 2e-3, 1e-4], np.float64),
                        **arrays, **{"pi/" + n: sample_idx[n] for n in names}, **{"ps/" + n: np.stack(samples[n]) for n in names},
                        **{"pg/" + n: np.stack(gsamples[n]) for n in names})
    print(f"wrote train_loop_{tag}: loss", out["loss"], "grad_norm", out["grad_norm"], "lr", out["lr"])


def _store_grads(named_grads, out, prefix, srng):
    """full tensors up to 4096 elements, else 1024 sampled elements + (sum, L2 norm) — the convention of train_grads_*.npz"""
    for name, g in named_grads:
        g = g.numpy().astype(F32).ravel()
        if g.size <= 4096:
            out[f"{prefix}g/" + name] = g
        else:
            idx = np.sort(srng.choice(g.size, size=1024, replace=False)).astype(np.int64)
            out[f"{prefix}gi/" + name] = idx
            out[f"{prefix}gs/" + name] = g[idx]
            out[f"{prefix}gn/" + name] = np.array([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum())])


def gen_aligner_grads(tag, B=5, seed=141):
    """torch autograd through the reference's PartAlignerTransformer (part_encoders.py:88-143): outputs and d (sum(mean dm) + sum(logvar dl)) / d (every
    parameter, part_code)."""
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = ref_import.build_reference_model("gen_chair.py", num_timesteps=10)
    load_all_weights(model)
    al = model.encoder.part_aligner
    al.train()
    rng = np.random.Generator(np.random.PCG64(seed))
    code = rng.standard_normal((B, 256, 4)).astype(F32)
    noise = rng.standard_normal((B, 32)).astype(F32)
    valid = np.ones((B, 4), F32)
    valid[1, 2] = 0
    valid[3, 0] = 0
    valid[3, 3] = 0
    dm, dl = rng.standard_normal((B, 3, 4)).astype(F32), rng.standard_normal((B, 3, 4)).astype(F32)
    for p_ in al.parameters():
        p_.grad = None
    z = torch.from_numpy(code).requires_grad_(True)
    mean, logvar = al(z, torch.from_numpy(valid), noise=torch.from_numpy(noise))
    ((mean * torch.from_numpy(dm)).sum() + (logvar * torch.from_numpy(dl)).sum()).backward()
    out = {"part_code": code, "noise": noise, "valid": valid, "d_mean": dm, "d_logvar": dl, "mean": mean.detach().numpy().astype(F32),
           "logvar": logvar.detach().numpy().astype(F32), "d_part_code": z.grad.numpy().astype(F32), "weight_seed": np.array(0)}
    _store_grads([(n, p_.grad) for n, p_ in al.named_parameters() if p_.grad is not None], out, "", np.random.Generator(np.random.PCG64(4244)))
    out["no_grad_params"] = np.array([n for n, p_ in al.named_parameters() if p_.grad is None])
    al.eval()
    np.savez_compressed(os.path.join(HERE, f"aligner_grads_{tag}.npz"), **out)
    print(f"wrote aligner_grads_{tag}:", len(out), "arrays; parameters without gradient:", list(out["no_grad_params"]), "| |mean| max", float(np.abs(out["mean"]).max()))


def gen_stage2_step(tag, B=4, N=64, T=10, seed=151):
    """One stage-2 training forward + backward of the reference's AnchorDiffAE (anchor_gen.py:970-1021 in train() mode, the gen_chair configuration =
    part aligner + fit_loss_type 4, every Dropout set to 0): the loss dict and d (sum of the 'loss' entries, runner.py:310-312 / parse_losses) / d (the
    aligner's parameters) — what stage 2 optimises (runner.py:76-90: the optimiser holds encoder.part_aligner.parameters()).  np.random is seeded for the
    timestep sampler (samplers/sampler.py:33); the torch draws (reparameterisation eps, the diffusion noise) are recorded."""
    with contextlib.redirect_stdout(io.StringIO()):
        model, cfg = ref_import.build_reference_model("gen_chair.py", num_timesteps=T)
    load_all_weights(model)
    model.npoints = N
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for p_ in model.parameters():
        p_.grad = None
    batch = make_batch(B, N, seed, absent=((2, 1),))
    batch["dp_present"] = batch["present"].copy()
    np.random.seed(seed)
    with DrawRecorder(seed + 1) as rec, contextlib.redirect_stdout(io.StringIO()):
        losses = model(to_torch(batch), device="cpu", epoch=3)
        total = sum(v.mean() for k, v in losses.items() if "loss" in k)
        total.backward()
    np.random.seed(seed)
    t = np.random.choice(T, size=(B,), p=np.ones(T) / T)
    out = {f"in/{k}": v for k, v in batch.items()}
    out.update(rec.as_dict())
    out.update({"loss/" + k: v.detach().numpy().astype(F32) for k, v in losses.items()})
    out.update({"total": np.array(float(total), F32), "t": t.astype(np.int64), "n_draws": np.array(len(rec.draws)), "np_seed": np.array(seed), "epoch": np.array(3),
                "weight_seed": np.array(0)})
    bn = model.encoder.encoder
    out.update({"bn/" + k: v.numpy().astype(F32) for k, v in bn.state_dict().items() if "running" in k})
    _store_grads([(n, p_.grad) for n, p_ in model.encoder.part_aligner.named_parameters() if p_.grad is not None], out, "", np.random.Generator(np.random.PCG64(4245)))
    model.eval()
    np.savez_compressed(os.path.join(HERE, f"stage2_step_{tag}.npz"), **out)
    print(f"wrote stage2_step_{tag}: total {float(total):.6f}", {k: float(v.detach().float().mean()) for k, v in losses.items() if 'loss' in k}, "t", t.tolist(), len(rec.draws), "draws",
          [a.shape for a in rec.draws])


def main():
    torch.manual_seed(0)
    if "--only-stage2" in sys.argv:
        gen_aligner_grads("B5")
        gen_stage2_step("B4_N64_T10")
        return
    if "--only-train-loop" not in sys.argv:
        gen_forward("gen_B2_K2_T10", gen=True)
        gen_forward("sample_B2_K2_T10", gen=False, seed=131)
        gen_encoder_forward("B3_N96")
    gen_train_loop("B3_N64_T10")
    gen_aligner_grads("B5")
    gen_stage2_step("B4_N64_T10")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own Python model on CPU (dev container only).

    python tests/golden/make_golden.py          # every fixture of THIS script (26 .npz, the two ddim_T40_* included)
    python tests/golden/make_golden.py --all    # + make_golden_pins.py + make_golden_sa.py (all 34) + MANIFEST.sha256 (manifest.py)

The reference (/root/reference) is imported through tools/ref_import.py (six stub modules,
SURVEY.md §8c).  Weights are NOT stored: both sides regenerate them with
difffacto_amd.synth.make_denoiser_weights(seed) and the reference loads them through
``load_state_dict``.  Fixtures hold inputs and the reference's outputs only (data, no code).

What is pinned
--------------
denoiser_eps_*.npz   TransformerNet.forward (attention.py:385) at chosen t
chain_T10_*.npz      AnchoredDiffusion.p_sample_loop_progressive (anchored_diffusion.py:528) for
                     T=10 with explicit noise (torch.randn / randn_like patched to replay the
                     recorded arrays, in the reference's own draw order), every step's sample,
                     and AnchorDiffAE.decode's dict (anchor_gen.py:145)
tables_T{10,100,1000}.npz  the schedule tables as the reference casts them to fp32
latents_*.npz        PartEncoder.sample_latents (part_encoders.py:1052-1110: flows in reverse, part
                     aligner, fixed_id mixing, K-fold repeat, seg-mask ids), torch.randn replayed; plus
                     one flow / the aligner called on their own
train_fwd_*.npz      AnchoredDiffusion.training_losses / q_sample (anchored_diffusion.py:760-853, :148-173) in eval mode
                     (dropout off), per-shape timesteps, with and without flags
train_grads_*.npz    autograd through TransformerNet in TRAIN mode with every Dropout set to p = 0 (the parity setting of
                     SURVEY.md §7 config 5): eps, the masked mse_loss of training_losses, and d loss / d (every parameter,
                     both ctx tensors).  Gradients of tensors with more than 4096 elements are stored as
                     (sum, L2 norm, 1024 elements at seeded positions) to keep the fixture small
prior_loss_*.npz     PartEncoder.get_prior_loss (part_encoders.py:1143-1182: flows forward + log-det, N(0, prior_var) log-likelihood,
                     posterior entropy, kl_weight) with autograd gradients for part_code, logvar and the 336 flow parameters
pointnet_v2_train_*.npz  the same class in TRAIN mode (batch-statistics BatchNorm): m, v, the running statistics it leaves behind, and
                     autograd gradients of sum(m dm) + sum(v dv) for every parameter (large tensors as sum / L2 norm / 1024 samples)
pointnet_v2_*.npz    PointNetV2.forward (pointnet.py:187-213, eval-mode BatchNorm), the encode-side part encoder
pn2_torch_*.npz      ball-query / grouping semantics from the reference's pure-torch PointNet++
                     (models/encoders/pointnet2_utils.py:84-104,41-57)
"""
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

F32 = np.float32


def load_denoiser_weights(model, W):
    sd = {k: torch.from_numpy(v.copy()) for k, v in W.items()}
    missing, unexpected = model.diffusion.model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def make_case(B, N, seed, all_valid):
    part_code, mean, logvar, valid = synth.make_latents(B, seed=seed, all_valid=all_valid)
    seg = synth.make_seg_mask(valid, N)
    return part_code, mean, logvar, valid, seg


def to_ref_inputs(part_code, mean, logvar, valid, seg):
    pc, m, lv, va = map(torch.from_numpy, (part_code, mean, logvar, valid))
    sg = torch.from_numpy(seg)
    from pointnet2_ops.pointnet2_utils import gather_operation
    anchors = gather_operation(m.contiguous(), sg)             # (B,3,N)  part_encoders.py:423
    logvar_pp = gather_operation(lv.contiguous(), sg)
    variance = torch.exp(logvar_pp)                            # anchor_gen.py:1044
    ctx = [pc, torch.cat([m, torch.exp(lv)], dim=1)]           # part_encoders.py:1317-1326 (log_scale_var = 0)
    return anchors, variance, ctx, va, sg


def gen_eps(model, tag, B, N, seed, all_valid, ts):
    case = make_case(B, N, seed, all_valid)
    anchors, variance, ctx, va, sg = to_ref_inputs(*case)
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    x = (np.sqrt(variance.numpy()) * rng.standard_normal((B, 3, N)).astype(F32) + anchors.numpy()).astype(F32)
    out = {}
    with torch.no_grad():
        for t in ts:
            tt = torch.tensor([t] * B)
            eps = model.diffusion.model(torch.from_numpy(x), tt, ctx, anchors=anchors.transpose(1, 2),
                                        variances=variance.transpose(1, 2), valid_id=va, anchor_assignment=sg)
            out[f"eps_t{t}"] = eps.numpy().astype(F32)
    np.savez_compressed(os.path.join(HERE, f"denoiser_eps_{tag}.npz"), x=x, part_code=case[0], mean=case[1],
                        logvar=case[2], valid=case[3], seg=case[4], ts=np.array(ts), weight_seed=np.array(0), **out)
    print("wrote denoiser_eps_" + tag, {k: float(np.abs(v).max()) for k, v in out.items()})


def gen_chain(model, tag, B, N, seed, all_valid, T, prefix="chain"):
    assert model.diffusion.num_timesteps == T
    T_exec = len(model.diffusion.steps)
    case = make_case(B, N, seed, all_valid)
    anchors, variance, ctx, va, sg = to_ref_inputs(*case)
    rng = np.random.Generator(np.random.PCG64(seed + 200))
    x_T_noise = rng.standard_normal((B, 3, N)).astype(F32)
    step_noise = rng.standard_normal((T_exec, B, 3, N)).astype(F32)
    queue = [torch.from_numpy(step_noise[i]) for i in range(T_exec)]
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def fake_randn(*shape, **kw):   # anchored_diffusion.py:564
        assert tuple(shape) == (B, 3, N)
        return torch.from_numpy(x_T_noise)

    def fake_randn_like(x):         # anchored_diffusion.py:476 (drawn every step incl. t=0)
        return queue.pop(0)

    samples, seq = {}, []
    try:
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        with torch.no_grad():
            for t, out in model.diffusion.p_sample_loop_progressive(
                    [B, 3, N], anchors=anchors, variance=variance, ctx=ctx, noise=None,
                    anchor_assignment=sg.to(torch.int32), valid_id=va, device="cpu"):
                samples[t] = out["sample"].numpy().astype(F32)
                seq.append(out["sample"].numpy().astype(F32))
        assert not queue
        # decode dict through the network-level entry point (anchor_gen.py:145-169)
        queue = [torch.from_numpy(step_noise[i]) for i in range(T_exec)]
        model.ret_traj, model.ret_interval = True, 5
        with torch.no_grad():
            dec = model.decode(anchors, ctx=ctx, variance=variance, anchor_assignments=sg.to(torch.int32),
                               valid_id=va, device="cpu")
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    order = [T] + list(model.diffusion.steps[::-1])            # executed t sequence (DDPM: T, T-1, ..., 0)
    if prefix == "chain":
        traj = np.stack([samples[t] for t in range(T, -1, -1)])   # index 0 = x_T, index T = x_0
    else:
        traj = np.stack(seq)                                    # execution order; 'quad' may visit t = 0 twice
    dec_np = {f"decode_{k}": v.numpy().astype(F32) for k, v in dec.items()}
    extra = {} if prefix == "chain" else {"steps": np.array(model.diffusion.steps), "ddim_eta": np.array(model.diffusion.ddim_eta, F32)}
    np.savez_compressed(os.path.join(HERE, f"{prefix}_T{T}_{tag}.npz"), **extra, part_code=case[0], mean=case[1], logvar=case[2],
                        valid=case[3], seg=case[4], x_T_noise=x_T_noise, step_noise=step_noise, traj=traj,
                        ret_interval=np.array(5), weight_seed=np.array(0), **dec_np)
    print(f"wrote {prefix}_T{T}_{tag}", traj.shape, float(np.abs(traj[-1]).max()), sorted(dec_np))


def load_latent_weights(model, W):
    sd = model.encoder.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v.copy())
    model.encoder.load_state_dict(sd, strict=True)


def gen_latents(model, tag, S, K, npoints, seed, fixed_id, all_valid):
    enc = model.encoder
    rng = np.random.Generator(np.random.PCG64(seed))
    w_noise = rng.standard_normal((S, enc.zdim, enc.n_class)).astype(F32)
    a_noise = rng.standard_normal((S * K, enc.part_aligner.noise_dim)).astype(F32)
    _, _, _, valid = synth.make_latents(S, seed=seed, all_valid=all_valid)
    real_randn = torch.randn
    queue = [w_noise, a_noise]

    def fake_randn(*shape, **kw):   # part_encoders.py:1054, :1065 (in this order)
        a = queue.pop(0)
        assert tuple(shape) == a.shape, (shape, a.shape)
        return torch.from_numpy(a)

    import contextlib
    import io
    try:
        torch.randn = fake_randn
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            ctx, mpp, lpp, seg, vid, lat = enc.sample_latents(S, npoints, "cpu", fixed_id=torch.tensor(fixed_id, dtype=torch.float32),
                                                              valid_id=torch.from_numpy(valid), epoch=0, K=K)
    finally:
        torch.randn = real_randn
    assert not queue
    part_code, mean, logvar, noise = lat
    with torch.no_grad():   # the two pieces on their own
        flow2 = enc.flow[2](torch.from_numpy(w_noise[..., 2].copy()), reverse=True).numpy().astype(F32)
        am, al = enc.part_aligner(torch.from_numpy(w_noise), torch.from_numpy(valid), noise=torch.from_numpy(a_noise[:S]))
    np.savez_compressed(os.path.join(HERE, f"latents_{tag}.npz"), w_noise=w_noise, aligner_noise=a_noise, valid_in=valid,
                        fixed_id=np.array(fixed_id, F32), K=np.array(K), npoints=np.array(npoints), weight_seed=np.array(0),
                        ctx0=ctx[0].numpy().astype(F32), ctx1=ctx[1].numpy().astype(F32), mean_per_point=mpp.numpy().astype(F32),
                        logvar_per_point=lpp.numpy().astype(F32), seg_mask=seg.numpy().astype(np.int32), valid_id=vid.numpy().astype(F32),
                        part_code=part_code.numpy().astype(F32), mean=mean.numpy().astype(F32), logvar=logvar.numpy().astype(F32),
                        noise=noise.numpy().astype(F32), flow2_reverse=flow2, aligner_mean=am.numpy().astype(F32),
                        aligner_logvar=al.numpy().astype(F32))
    print("wrote latents_" + tag, float(np.abs(part_code.numpy()).max()), float(np.abs(mean.numpy()).max()),
          float(np.abs(logvar.numpy()).max()))


def gen_pointnet_v2(model, tag, B, N, seed):
    """PointNetV2.forward (pointnet.py:187-213) in eval mode with randomised BN statistics."""
    enc = model.encoder.encoder
    rng = np.random.Generator(np.random.PCG64(seed))
    W = synth.make_pointnet_v2_weights(seed=0)
    sd = enc.state_dict()
    assert {k for k in sd if not k.endswith("num_batches_tracked")} == set(W)
    for k, a in W.items():
        assert tuple(sd[k].shape) == a.shape, k
        sd[k] = torch.from_numpy(a.copy())
    enc.load_state_dict(sd)
    enc.eval()
    x = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    seg = rng.integers(0, 4, size=(B, N))
    seg[0][seg[0] == 3] = 0                      # shape 0 has no part 3: its pooled features are max(0 * x) = 0
    attn = np.eye(4, dtype=F32)[seg]
    with torch.no_grad():
        m, v = enc(torch.from_numpy(x), torch.from_numpy(attn))
    np.savez_compressed(os.path.join(HERE, f"pointnet_v2_{tag}.npz"), x=x, attn=attn, m=m.numpy().astype(F32), v=v.numpy().astype(F32),
                        weight_seed=np.array(0))
    print("wrote pointnet_v2_" + tag, m.shape, float(m.abs().max()))


def gen_training_losses(model, tag, B, N, seed, T):
    """AnchoredDiffusion.training_losses (anchored_diffusion.py:760-853) in eval mode (dropout off), per-shape t."""
    case = make_case(B, N, seed, False)
    anchors, variance, ctx, va, sg = to_ref_inputs(*case)
    rng = np.random.Generator(np.random.PCG64(seed + 300))
    x0 = (np.sqrt(variance.numpy()) * rng.standard_normal((B, 3, N)).astype(F32) * 0.5 + anchors.numpy()).astype(F32)
    noise = rng.standard_normal((B, 3, N)).astype(F32)
    t = rng.integers(0, T, size=(B,)).astype(np.int64)
    flags = (rng.uniform(size=(B, 1, N)) > 0.2).astype(F32)
    out = {}
    with torch.no_grad():
        for name, fl in (("flags", torch.from_numpy(flags)), ("noflags", None)):
            r = model.diffusion.training_losses(torch.from_numpy(x0), torch.from_numpy(t), anchors=anchors, variance=variance, ctx=ctx,
                                                anchor_assignment=sg.to(torch.int32), valid_id=va, flags=fl, noise=torch.from_numpy(noise))
            out["mse_loss_" + name] = np.array(float(r["mse_loss"]), F32)
        x_t = model.diffusion.q_sample(torch.from_numpy(x0), torch.from_numpy(t), anchors, noise=torch.from_numpy(noise), variance=variance)
    np.savez_compressed(os.path.join(HERE, f"train_fwd_{tag}.npz"), part_code=case[0], mean=case[1], logvar=case[2], valid=case[3], seg=case[4],
                        x_start=x0, noise=noise, t=t, flags=flags, x_t=x_t.numpy().astype(F32), weight_seed=np.array(0), **out)
    print("wrote train_fwd_" + tag, out)


def gen_train_grads(model, tag, B, N, seed, T):
    """loss.backward() through the reference's TransformerNet (attention.py:385-440) and mse_loss (anchored_diffusion.py:840-847)."""
    case = make_case(B, N, seed, False)
    anchors, variance, ctx, va, sg = to_ref_inputs(*case)
    rng = np.random.Generator(np.random.PCG64(seed + 300))
    x0 = (np.sqrt(variance.numpy()) * rng.standard_normal((B, 3, N)).astype(F32) * 0.5 + anchors.numpy()).astype(F32)
    noise = rng.standard_normal((B, 3, N)).astype(F32)
    t = rng.integers(0, T, size=(B,)).astype(np.int64)
    flags = (rng.uniform(size=(B, 1, N)) > 0.2).astype(F32)
    net = model.diffusion.model
    model.train()
    ndrop = 0
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
            ndrop += 1
    for p in net.parameters():
        p.grad = None
    ctx = [c.clone().requires_grad_(True) for c in ctx]
    r = model.diffusion.training_losses(torch.from_numpy(x0), torch.from_numpy(t), anchors=anchors, variance=variance, ctx=ctx,
                                        anchor_assignment=sg.to(torch.int32), valid_id=va, flags=torch.from_numpy(flags),
                                        noise=torch.from_numpy(noise))
    loss = r["mse_loss"]
    loss.backward()
    with torch.no_grad():
        x_t = model.diffusion.q_sample(torch.from_numpy(x0), torch.from_numpy(t), anchors, noise=torch.from_numpy(noise), variance=variance)
        eps = net(x_t, torch.from_numpy(t), [c.detach() for c in ctx], anchors=anchors.transpose(1, 2), variances=variance.transpose(1, 2),
                  valid_id=va, anchor_assignment=sg)
    out = {"loss": np.array(float(loss), F32), "eps": eps.numpy().astype(F32), "x_t": x_t.numpy().astype(F32),
           "d_ctx_code": ctx[0].grad.numpy().astype(F32), "d_ctx_mv": ctx[1].grad.numpy().astype(F32)}
    srng = np.random.Generator(np.random.PCG64(4242))
    for name, p in net.named_parameters():
        g = p.grad.numpy().astype(F32).ravel()
        if g.size <= 4096:
            out["g/" + name] = g
        else:
            idx = np.sort(srng.choice(g.size, size=1024, replace=False)).astype(np.int64)
            out["gi/" + name] = idx
            out["gs/" + name] = g[idx]
            out["gn/" + name] = np.array([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum())])
    model.eval()
    np.savez_compressed(os.path.join(HERE, f"train_grads_{tag}.npz"), part_code=case[0], mean=case[1], logvar=case[2], valid=case[3],
                        seg=case[4], x_start=x0, noise=noise, t=t, flags=flags, weight_seed=np.array(0), **out)
    print("wrote train_grads_" + tag, float(loss), ndrop, "dropouts zeroed;", len(out), "arrays")


def gen_pointnet_v2_train(model, tag, B, N, seed):
    """PointNetV2 (pointnet.py:187-213) in train() mode + loss.backward(), from the reference class itself."""
    enc = model.encoder.encoder
    rng = np.random.Generator(np.random.PCG64(seed))
    W = synth.make_pointnet_v2_weights(seed=0)
    sd = enc.state_dict()
    for k, a in W.items():
        sd[k] = torch.from_numpy(a.copy())
    enc.load_state_dict(sd)
    enc.train()
    for p in enc.parameters():
        p.grad = None
    x = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    seg = rng.integers(0, 4, size=(B, N))
    seg[0][seg[0] == 3] = 0
    attn = np.eye(4, dtype=F32)[seg]
    dm = rng.standard_normal((B, 4, 256)).astype(F32)
    dv = rng.standard_normal((B, 4, 256)).astype(F32)
    m, v = enc(torch.from_numpy(x), torch.from_numpy(attn))
    ((m * torch.from_numpy(dm)).sum() + (v * torch.from_numpy(dv)).sum()).backward()
    out = {"m": m.detach().numpy().astype(F32), "v": v.detach().numpy().astype(F32)}
    for k, t in enc.state_dict().items():
        if "running" in k:
            out["r/" + k] = t.numpy().astype(F32)
    srng = np.random.Generator(np.random.PCG64(4243))
    for name, p in enc.named_parameters():
        g = p.grad.numpy().astype(F32).ravel()
        if g.size <= 4096:
            out["g/" + name] = g
        else:
            idx = np.sort(srng.choice(g.size, size=1024, replace=False)).astype(np.int64)
            out["gi/" + name], out["gs/" + name] = idx, g[idx]
            out["gn/" + name] = np.array([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum())])
    enc.eval()
    np.savez_compressed(os.path.join(HERE, f"pointnet_v2_train_{tag}.npz"), x=x, attn=attn, dm=dm, dv=dv, weight_seed=np.array(0), **out)
    print("wrote pointnet_v2_train_" + tag, len(out), "arrays", float(np.abs(out["m"]).max()))


def gen_prior_loss(model, tag, B, seed):
    """get_prior_loss + backward from the reference encoder itself (kl_weight / prior_var as built by ref_import)."""
    enc = model.encoder
    rng = np.random.Generator(np.random.PCG64(seed))
    part_code = rng.standard_normal((B, 256, 4)).astype(F32)
    mean = rng.standard_normal((B, 4, 256)).astype(F32)
    logvar = (-1.0 + 0.3 * rng.standard_normal((B, 4, 256))).astype(F32)
    pats = np.array([[1, 1, 1, 1], [1, 1, 1, 0], [0, 1, 1, 0], [1, 1, 0, 0], [0, 1, 1, 1]], dtype=F32)
    valid = pats[np.arange(B) % len(pats)]
    for p in enc.parameters():
        p.grad = None
    z = torch.from_numpy(part_code).requires_grad_(True)
    lv = torch.from_numpy(logvar).requires_grad_(True)
    old_kl, enc.kl_weight = enc.kl_weight, 5e-4        # configs/train_chair_stage1.py:13 (the gen config carries 0)
    # loss_dict['kl_weight'] = torch.ones(1).cuda() * kl_weight: no CUDA here
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        d = enc.get_prior_loss(z, torch.from_numpy(mean), lv, torch.from_numpy(valid), epoch=0)
    finally:
        del torch.Tensor.cuda
        enc.kl_weight = old_kl
    d["prior_loss"].backward()
    out = {"loss": np.array(float(d["prior_loss"]), F32), "kl_weight": np.array(float(d["kl_weight"]), F32), "prior_var": np.array(float(enc.prior_var), F32),
           "d_part_code": z.grad.numpy().astype(F32), "d_logvar": lv.grad.numpy().astype(F32),
           "log_p_mean": np.array([float(d[f"log_p_part_{i}"]) for i in range(4)], F32),
           "entropy_mean": np.array([float(d[f"entropy_{i}"]) for i in range(4)], F32)}
    srng = np.random.Generator(np.random.PCG64(4244))
    for name, p in enc.named_parameters():
        if not name.startswith("flow.") or p.grad is None:
            continue
        g = p.grad.numpy().astype(F32).ravel()
        if g.size <= 512:
            out["g/" + name] = g
        else:
            idx = np.sort(srng.choice(g.size, size=64, replace=False)).astype(np.int64)
            out["gi/" + name], out["gs/" + name] = idx, g[idx]
            out["gn/" + name] = np.array([g.astype(np.float64).sum(), np.sqrt((g.astype(np.float64) ** 2).sum())])
    np.savez_compressed(os.path.join(HERE, f"prior_loss_{tag}.npz"), part_code=part_code, mean=mean, logvar=logvar, valid=valid,
                        weight_seed=np.array(0), **out)
    print("wrote prior_loss_" + tag, float(d["prior_loss"]), float(d["kl_weight"]), len(out), "arrays")


def gen_tables():
    from difffacto.models.diffusions.diffusion_utils import extract_into_tensor
    from difffacto.utils.registry import DIFFUSIONS
    names = ["sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_mean_coef1", "posterior_mean_coef2", "posterior_mean_coef3",
             "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"]
    for T in (10, 100, 1000):
        model, _ = ref_import.build_reference_model(num_timesteps=T)
        d = model.diffusion
        t = torch.arange(T)
        out = {n: extract_into_tensor(getattr(d, n), t, (T,)).numpy().astype(F32) for n in names}
        np.savez_compressed(os.path.join(HERE, f"tables_T{T}.npz"), **out)
        print("wrote tables", T)


def gen_pn2_torch():
    """Reference pure-torch PointNet++ helpers as a semantic cross-check for ball query/grouping."""
    ref_import.import_reference()
    from difffacto.models.encoders.pointnet2_utils import query_ball_point, index_points
    rng = np.random.Generator(np.random.PCG64(7))
    B, N, M, ns, r = 2, 256, 32, 16, 0.35
    xyz = rng.uniform(-1, 1, size=(B, N, 3)).astype(F32)
    new_xyz = xyz[:, rng.permutation(N)[:M]].copy()
    idx = query_ball_point(r, ns, torch.from_numpy(xyz), torch.from_numpy(new_xyz))
    grouped = index_points(torch.from_numpy(xyz), idx)
    np.savez_compressed(os.path.join(HERE, "pn2_torch_ballquery.npz"), xyz=xyz, new_xyz=new_xyz, radius=np.array(r, F32),
                        nsample=np.array(ns), idx=idx.numpy().astype(np.int64), grouped=grouped.numpy().astype(F32))
    print("wrote pn2_torch_ballquery")


def main():
    torch.manual_seed(0)
    W = synth.make_denoiser_weights(seed=0)
    model, _ = ref_import.build_reference_model(num_timesteps=10)
    load_denoiser_weights(model, W)
    if "--only-pn2" in sys.argv:
        gen_pn2_torch()
        return
    load_latent_weights(model, synth.make_latent_weights(seed=0))
    gen_latents(model, "S3_K2_mixed", S=3, K=2, npoints=64, seed=31, fixed_id=[0, 0, 0, 0], all_valid=False)
    gen_latents(model, "S4_K3_fixed", S=4, K=3, npoints=32, seed=32, fixed_id=[0, 1, 0, 0], all_valid=False)
    gen_pointnet_v2(model, "B3_N200", B=3, N=200, seed=51)
    gen_training_losses(model, "B3_N64_T10", B=3, N=64, seed=71, T=10)
    if "--only-prior" in sys.argv:
        gen_prior_loss(model, "B6", B=6, seed=95)
        return
    if "--only-pnv2-train" in sys.argv:
        gen_pointnet_v2_train(model, "B5_N160", B=5, N=160, seed=91)
        return
    gen_train_grads(model, "B3_N64_T10", B=3, N=64, seed=81, T=10)
    gen_pointnet_v2_train(model, "B5_N160", B=5, N=160, seed=91)
    gen_prior_loss(model, "B6", B=6, seed=95)
    if "--only-train" in sys.argv:
        return
    if True:   # the DDIM fixtures (also what --only-ddim / --only-latents stop after): the flag-less run regenerates EVERY fixture of this script
        from difffacto.config.config import get_cfg
        from difffacto.utils.registry import build_from_cfg, MODELS
        cfg = get_cfg()
        for name, kw in (("quad8_eta1", dict(ddim_nsteps=8, ddim_discretize="quad", ddim_eta=1.0)),
                         ("uniform5_eta0", dict(ddim_nsteps=5, ddim_discretize="uniform", ddim_eta=0.0))):
            cfg.model["num_timesteps"] = 40
            cfg.model["diffusion"].update(ddim_sampling=True, **kw)
            dm = build_from_cfg(cfg.model, MODELS).eval()
            load_denoiser_weights(dm, W)
            gen_chain(dm, name + "_B2_N64", B=2, N=64, seed=61, all_valid=False, T=40, prefix="ddim")
        cfg.model["diffusion"].update(ddim_sampling=False)
    if "--only-latents" in sys.argv or "--only-ddim" in sys.argv:
        return
    gen_eps(model, "B2_N128_mixed", B=2, N=128, seed=11, all_valid=False, ts=[0, 3, 9])
    gen_eps(model, "B2_N128_allvalid", B=2, N=128, seed=12, all_valid=True, ts=[5])
    gen_eps(model, "B1_N2048", B=1, N=2048, seed=13, all_valid=True, ts=[7])
    gen_chain(model, "B2_N128_mixed", B=2, N=128, seed=21, all_valid=False, T=10)
    gen_chain(model, "B3_N64_allvalid", B=3, N=64, seed=22, all_valid=True, T=10)
    gen_tables()
    gen_pn2_torch()
    if "--all" in sys.argv:   # the other two generators + the manifest: one command regenerates all fixtures of tests/golden/
        import make_golden_pins
        import make_golden_sa
        import manifest
        make_golden_pins.main()
        make_golden_sa.main()
        manifest.write()


if __name__ == "__main__":
    main()

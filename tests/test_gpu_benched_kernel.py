"""Oracle-side evidence for the kernels bench.py actually times: `k_denoise_pipe<8>` (bf16, the headline line) and
`k_denoise_pipe_f32<8>` (the `f32` block), at the benched chain length and at a batch that needs more than one round of workgroups
per CU.  Every launch is forced to its variant (include/dfx_debug.h) and the variant that RAN is asserted
(dfx_last_kernel_variant, include/dfx.h).

  * long-chain variant equivalence: T = 1000, explicit noise, the benched kernel against the co-operative kernel (the one the
    reference goldens and the T = 1000 oracle tests ran through in earlier rounds) and the 4 / 2-wavefront and two-tile variants,
    bit for bit; at B = 3 (one partial round, natural workgroup order) and at B = 44 (352 workgroups: two rounds on 256 CUs, five
    XCD-remap periods + a tail of 4 shapes in natural order), where three shapes of the big launch are compared with a 3-shape launch;
  * teacher-forced sweep on the plain random-init weights (full-scale eps, no contraction trick): the PyTorch-CPU oracle
    (oracle/torch_cpu.py, pinned to the reference goldens; anchored_diffusion.py:450-484, attention.py:385-440) walks 2 x 256 points
    through all 1000 steps; at EVERY t the oracle's own x_t goes through dfx_denoise_eps and dfx_p_sample of the forced kernel and
    eps, x_{t-1} and pred_xstart are gated — no chaos, because the chain is never fed its own output.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from difffacto_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _engine(W, T, prec):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.engine import DenoiserEngine
    return DenoiserEngine({k: torch.from_numpy(v) for k, v in W.items()}, num_timesteps=T, precision=prec)


def _ctx(eng, lat, idx=None):
    pc, mean, var, valid = lat
    sel = (lambda a: a) if idx is None else (lambda a: np.ascontiguousarray(a[idx]))
    return eng.prepare_shapes(*(torch.from_numpy(sel(a)) for a in (pc, mean, var, valid)))


@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_T1000_chain_benched_kernel_is_bit_identical_to_every_other_variant(prec):
    from _variants import forced, ran
    T, N = 1000, 2048
    W = synth.make_denoiser_weights(seed=0)
    eng = _engine(W, T, prec)
    g = torch.Generator(device="cuda").manual_seed(17)
    others = (4, 2, 1, 64, 16) if prec == "bf16" else (4, 2, 1)     # 1 = co-operative kernel (bf16) / direct kernel (fp32); 64 = pipe2; 16 = co-operative, two tiles per workgroup
    # ---- B = 3: every variant on the whole launch ----
    B = 3
    pc, mean, logvar, valid = synth.make_latents(B, seed=7)
    lat = (pc, mean, np.exp(logvar).astype(np.float32), valid)
    seg = torch.from_numpy(synth.make_seg_mask(valid, N))
    xT = torch.randn(B, 3, N, device="cuda", generator=g)
    zs = torch.randn(T, B, 3, N, device="cuda", generator=g)
    cx = _ctx(eng, lat)
    with forced(8):
        ref, traj = eng.sample_chain(cx, seg, x_T_noise=xT, step_noise=zs, ret_interval=100)
        name8 = ran(prec, 8)
    assert torch.isfinite(ref).all()
    for nw in others:
        with forced(nw):
            o, tr = eng.sample_chain(cx, seg, x_T_noise=xT, step_noise=zs, ret_interval=100)
            name = ran(prec, nw)
        assert torch.equal(o, ref) and torch.equal(tr, traj), (name, (o - ref).abs().max().item())
        print(f"T={T} B={B}: {name} == {name8} bit for bit (cloud + 10 trajectory snapshots)")
    del zs
    # ---- B = 44: > 1 round of workgroups per CU, XCD remap (5 periods of 8 shapes) + natural-order tail (4 shapes) ----
    B = 44
    pc, mean, logvar, valid = synth.make_latents(B, seed=8)
    lat = (pc, mean, np.exp(logvar).astype(np.float32), valid)
    segn = synth.make_seg_mask(valid, N)
    xT = torch.randn(B, 3, N, device="cuda", generator=g)
    zs = torch.randn(T, B, 3, N, device="cuda", generator=g)
    with forced(8):
        big, _ = eng.sample_chain(_ctx(eng, lat), torch.from_numpy(segn), x_T_noise=xT, step_noise=zs)
        ran(prec, 8)
    assert torch.isfinite(big).all()
    idx = np.array([5, 26, 42])          # remapped period 0, remapped period 3, natural-order tail
    sub_seg = torch.from_numpy(np.ascontiguousarray(segn[idx]))
    sub_xT, sub_zs = xT[idx].contiguous(), zs[:, idx].contiguous()
    for nw in (1, 2):
        with forced(nw):
            small, _ = eng.sample_chain(_ctx(eng, lat, idx), sub_seg, x_T_noise=sub_xT, step_noise=sub_zs)
            name = ran(prec, nw)
        assert torch.equal(small, big[idx]), (name, (small - big[idx]).abs().max().item())
        print(f"T={T}: shapes {idx.tolist()} of the B={B} launch of {name8} (352 workgroups) == a 3-shape launch of {name}")
    # the Philox path of the big launch: same bits from a 3-shape launch with the shapes' global offsets
    with forced(8):
        bigp, _ = eng.sample_chain(_ctx(eng, lat), torch.from_numpy(segn), seed=3)
    for i in idx:
        with forced(1):
            one, _ = eng.sample_chain(_ctx(eng, lat, [int(i)]), torch.from_numpy(np.ascontiguousarray(segn[[i]])), seed=3, shape_offset=int(i))
        assert torch.equal(one[0], bigp[i]), int(i)
    eng.close()


# measured on MI355X (profiles/r04_parity_prints.txt), largest value over the 1000 steps, relative to max(1, |ref|max) at that t:
#   bf16  eps 2.68e-3 (t=11)    x_(t-1) 1.51e-5 (t=997)   pred_xstart 7.8e-4 (t=997)   -> gates <= 3x (GPU-side bf16 rounding dominates; before the
#         bias fold of round 4 — b1' on the constant-one K slot, DESIGN 5.1 — the same run gave 2.15e-3 / 1.16e-5 / 5.8e-4)
#   f32   eps 2.7e-6  (t=336)   x_(t-1) 1.3e-7  (t=970)   pred_xstart 8.7e-7 (t=997)   -> gates at ~10x: this is fp32-vs-fp32 against a CPU
#         oracle whose own summation order depends on the host's BLAS and thread count
TEACHER_GATES = {"f32": dict(eps=3e-5, x=1.5e-6, x0=1e-5), "bf16": dict(eps=6.4e-3, x=3.5e-5, x0=1.75e-3)}


@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_teacher_forced_every_t_of_T1000_random_init_vs_cpu_oracle(prec):
    from _variants import forced, ran
    from oracle import diffusion as odf
    from oracle import torch_cpu as tc
    T, B, N = 1000, 2, 256
    W = synth.make_denoiser_weights(seed=0)          # plain random-init: eps_theta is O(1) at every t
    pc, mean, logvar, valid = synth.make_latents(B, seed=5)
    valid[1] = np.array([1, 0, 1, 1], np.float32)    # one shape with an absent part: the mask path at every t
    var = np.exp(logvar).astype(np.float32)
    seg = synth.make_seg_mask(valid, N)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    anchors, variance = odf.gather_params(seg, mean, var)
    ctx = [tt(pc), tt(np.concatenate([mean, var], 1))]
    Wt = {k: tt(v) for k, v in W.items()}
    tb = odf.Tables(T)
    g = torch.Generator().manual_seed(23)
    xT = torch.randn(B, 3, N, generator=g)
    zs = torch.randn(T, B, 3, N, generator=g)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    xs, eps_o, x0_o = [], [], []
    with torch.no_grad():
        x = torch.sqrt(tt(variance)) * xT + tt(anchors)
        for i, t in enumerate(range(T - 1, -1, -1)):
            xs.append(x)
            x, e = tc.p_sample(tb, Wt, x, t, tt(anchors), ctx, tt(variance), tt(seg), tt(valid), zs[i])
            eps_o.append(e)
            f = lambda name: float(getattr(tb, name).astype("float32")[t])
            x0_o.append(f("sqrt_recip_alphas_cumprod") * (xs[-1] - tt(anchors)) + tt(anchors)
                        - f("sqrt_recipm1_alphas_cumprod") * torch.sqrt(tt(variance)) * e)
        xs.append(x)
    assert torch.isfinite(x).all()
    X = torch.stack(xs).cuda()                       # X[i] = x_t at t = T-1-i; X[i+1] = the oracle's x_{t-1}
    E, X0, Z = torch.stack(eps_o).cuda(), torch.stack(x0_o).cuda(), zs.cuda()
    eng = _engine(W, T, prec)
    cx = eng.prepare_shapes(tt(pc), tt(mean), tt(var), tt(valid))
    sg = tt(seg)
    worst = dict(eps=0.0, x=0.0, x0=0.0)
    at = dict(eps=-1, x=-1, x0=-1)
    scale = dict(eps=0.0, x=0.0, x0=0.0)
    with forced(8):
        for i, t in enumerate(range(T - 1, -1, -1)):
            e = eng.eps(cx, X[i], sg, t)
            ran(prec, 8)
            xp, x0 = eng.p_sample(cx, X[i], sg, t, noise=Z[i], want_xstart=True)
            ran(prec, 8)
            # deviations relative to the size of the quantity at this t (eps is O(1); x and pred_xstart grow with the cloud)
            for k, got, ref in (("eps", e, E[i]), ("x", xp, X[i + 1]), ("x0", x0, X0[i])):
                s = max(1.0, float(ref.abs().max()))
                d = float((got - ref).abs().max()) / s
                scale[k] = max(scale[k], s)
                if d > worst[k]:
                    worst[k], at[k] = d, t
    eng.close()
    print(f"teacher-forced T={T} random-init [{prec}, k_denoise_pipe{'_f32' if prec == 'f32' else ''}<8>] max over all t of max-abs / max(1, |ref|max): "
          f"eps {worst['eps']:.3e} (t={at['eps']}), x_(t-1) {worst['x']:.3e} (t={at['x']}), pred_xstart {worst['x0']:.3e} (t={at['x0']}); "
          f"largest |ref|: eps {scale['eps']:.2f}, x {scale['x']:.1f}, x0 {scale['x0']:.1f}")
    gate = TEACHER_GATES[prec]
    assert worst["eps"] < gate["eps"] and worst["x"] < gate["x"] and worst["x0"] < gate["x0"], worst

#!/usr/bin/env python
"""Stage 2 of the reference's training recipe (configs/gen_chair.py with train_aligner=True / train_*_stage2.py): only the part aligner is optimised
(runner/runner.py:76-90), against fit_loss (type 4) + the diffusion MSE, with cIMLE noise caching every `cache_interval` epochs (runner.py:148-170).

    python examples/train_stage2.py [--batch 32] [--npoints 2048] [--iters 20]

Synthetic data (there is no dataset here): parts are Gaussians around `part_shift` with std `part_scale`.  Everything heavy runs on libdfx through the
mirrors of the reference's classes (difffacto_amd/networks.py); the loop is the reference's: cache_noise -> zero_grad -> loss dict -> backward -> clip + Adam."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difffacto_amd import synth, training  # noqa: E402
from difffacto_amd.networks import AnchorDiffAE  # noqa: E402

ENC = dict(type="PartEncoderForTransformerDecoder", encoder=dict(type="PointNetV2", zdim=256, point_dim=3, per_part_mlp=True),
           part_aligner=dict(type="PartAlignerTransformer", in_channels=256, out_channels=6, n_class=4, d_head=32, depth=5, n_heads=8, dropout=0., use_linear=True,
                             class_cond=True, single_attn=True, add_class_cond=True, cimle=True, noise_scale=100, cond_noise_type=0),
           n_class=4, kl_weight=0, fit_loss_type=4, fit_loss_weight=1.0, use_flow=True, latent_flow_depth=14, latent_flow_hidden_dim=256, include_z=False,
           include_part_code=True, include_params=True, use_gt_params=False, gen=True, prior_var=1.0)
NET = dict(type="TransformerNet", in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=0.2, context_dim=262, n_class=4, class_cond=True,
           use_linear=True, cat_params_to_x=True, single_attn=True, cat_class_to_x=True)
DIFF = dict(type="AnchoredDiffusion", net=NET, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode="linear", use_beta=False, model_mean_type="epsilon",
            learn_variance=True, loss_type="mse", include_anchors=False)


def batch(B, N, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    _, shift, lv, valid = synth.make_latents(B, seed=seed)
    seg = synth.make_seg_mask(valid, N)
    std = np.exp(0.5 * lv).astype(np.float32)
    idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
    pts = (np.take_along_axis(shift, idx, 2) + np.take_along_axis(std, idx, 2) * rng.standard_normal((B, 3, N))).astype(np.float32).transpose(0, 2, 1)
    attn = np.eye(4, dtype=np.float32)[seg]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return {"input": t(pts), "ref": t(pts), "seg_mask": t(seg.astype(np.int64)), "ref_seg_mask": t(seg.astype(np.int64)), "attn_map": t(attn), "ref_attn_map": t(attn),
            "present": t(valid), "dp_present": t(valid), "part_shift": t(shift), "part_scale": t(std), "noise": torch.zeros(B, 32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    model = AnchorDiffAE(encoder=ENC, diffusion=DIFF, sampler=dict(type="Uniform"), num_anchors=4, num_timesteps=1000, npoints=a.npoints, gen=True, cimle=True,
                         cimle_sample_num=1, sample_noise_num=20).cuda()
    for n, p in model.named_parameters():                       # stage 2 optimises the aligner only (runner.py:88): the rest needs no backward kernels
        p.requires_grad_(n.startswith("encoder.part_aligner."))
    opt = training.Adam(list(model.encoder.part_aligner.parameters()), lr=2e-3, max_norm=10.0)
    pcds = batch(a.batch, a.npoints, 0)
    model.eval()
    pcds["noise"] = model.cache_noise(pcds, "cuda").cpu()       # the best of 20 aligner noises per shape (anchor_gen.py:807-815)
    model.train()
    t0 = time.perf_counter()
    for it in range(a.iters):
        opt.zero_grad()
        losses = model(pcds, device="cuda", epoch=it)
        total = sum(v.mean() for k, v in losses.items() if "loss" in k)          # parse_losses (utils/misc.py:120-132)
        total.backward()
        opt.step()
        if it % 5 == 0 or it == a.iters - 1:
            print(f"iter {it:3d}  total {float(total.detach()):.4f}  fit {float(losses['fit_loss'].mean()):.4f}  mse {float(losses['mse_loss']):.4f}")
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / a.iters * 1e3:.1f} ms per stage-2 iteration (B = {a.batch} x {a.npoints} points, denoiser with dropout 0.2, bf16 products)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Stage-1 training loop (configs/train_chair_stage1.py: PointNetV2 part encoder + per-part latent flows + cross-diffusion
denoiser, ground-truth anchors) on the HIP path, shaped like the reference's Runner.train (runner.py:299-316):

    losses = model(pcds)           -> training.stage1_losses: encoder.forward (PointNetV2 train kernels, prior loss) + denoiser MSE
    sum of '*loss*' -> backward    -> libdfx backward kernels (denoiser, PointNetV2, flows)
    clip_grad_norm_(10) + Adam     -> training.Adam (one launch each per flat gradient buffer: denoiser, PointNetV2, flows)

on synthetic shapes (no data set in this image).   python examples/train_stage1.py --iters 10 --batch 32
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import synth, training
from difffacto_amd.encoders import PartEncoderForTransformerDecoder
from difffacto_amd.modules import AnchoredDiffusion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"], help="matrix products of the denoiser")
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--encoder-precision", default="f32", choices=["bf16", "f32"], help="matrix products of the PointNetV2 trunk")
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--two-pass-bn", action="store_true", help="A/B: BatchNorm batch statistics with two passes over each layer's output")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.two_pass_bn:
        from difffacto_amd import _ffi
        _ffi.lib().dfx_debug_bn_fused_stats(0)
    enc = PartEncoderForTransformerDecoder(encoder=dict(type="PointNetV2", zdim=256, per_part_mlp=True), n_class=4, part_aligner=None,
                                           include_z=False, include_part_code=True, include_params=True, use_gt_params=True, kl_weight=5e-4,
                                           use_flow=True, latent_flow_depth=14, latent_flow_hidden_dim=256, gen=True, prior_var=1.0)
    net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=a.dropout, context_dim=256 + 6,
               n_class=4, class_cond=True, use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True, cat_class_to_x=True)
    diff = AnchoredDiffusion(net=net, num_timesteps=a.timesteps, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False,
                             rescale_timesteps=False, model_mean_type="epsilon", learn_variance=True, loss_type='mse', include_anchors=False,
                             precision=a.precision)
    enc.encoder.train_precision = a.encoder_precision
    enc, diff = enc.cuda().train(), diff.cuda().train()
    opt = training.Adam(list(enc.parameters()) + list(diff.model.parameters()), lr=a.lr, max_norm=10.0)   # clip over everything
    rng = np.random.Generator(np.random.PCG64(0))
    B, N = a.batch, a.npoints
    cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    for it in range(a.iters):
        _, shift, lv, valid = synth.make_latents(B, seed=it)
        seg = synth.make_seg_mask(valid, N)
        std = np.exp(0.5 * lv).astype(np.float32)
        idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
        pts = (np.take_along_axis(shift, idx, 2) + np.take_along_axis(std, idx, 2) * rng.standard_normal((B, 3, N))).astype(np.float32)
        pcds = {"input": cu(pts.transpose(0, 2, 1)), "ref": cu(pts.transpose(0, 2, 1)), "present": cu(valid), "dp_present": cu(valid),
                "ref_seg_mask": cu(seg.astype(np.int64)), "ref_attn_map": cu(np.eye(4, dtype=np.float32)[seg]), "part_shift": cu(shift),
                "part_scale": cu(std), "noise": torch.zeros(B, 32).cuda()}
        t0 = time.perf_counter()
        opt.zero_grad()
        losses = training.stage1_losses(enc, diff, pcds, epoch=it)
        total = sum(v.sum() for k, v in losses.items() if "loss" in k)                     # parse_losses, misc.py:120-132
        total.backward()
        norm = opt.step()                                                                 # clip_grad_norm_(10) + Adam, one launch per gradient buffer
        torch.cuda.synchronize()
        print(f"iter {it}: prior_loss {float(losses['prior_loss'].detach()):.3f}  mse_loss {float(losses['mse_loss'].detach()):.4f}  "
              f"grad norm {float(norm):.2f}  {(time.perf_counter() - t0) * 1e3:.1f} ms ({B} shapes)", flush=True)


if __name__ == "__main__":
    main()

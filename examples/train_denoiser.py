#!/usr/bin/env python
"""Training loop of the cross-diffusion denoiser on the HIP path, shaped like the reference's Runner.train
(python/difffacto/runner/runner.py:299-316) for `train_chair_stage1.py`'s denoiser objective:

    t ~ Uniform{0..T-1} per shape -> x_t = q_sample(x0, t) -> eps_hat = TransformerNet(x_t, t, ctx) -> masked MSE
    -> backward -> (DDP: one flat-bucket gradient all-reduce) -> clip_grad_norm_(10) -> Adam(lr by LinearLR)

on synthetic shapes (there is no dataset in this image): part latents / anchors from difffacto_amd.synth, x0 = anchor +
sqrt(variance) * noise.  One process per GPU:

    python examples/train_denoiser.py --iters 20 --batch 32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_denoiser.py --iters 20
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from difffacto_amd import parallel, synth, training
from difffacto_amd.modules import AnchoredDiffusion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="shapes per GPU")
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--dropout", type=float, default=0.0, help="nn.Dropout p of the denoiser (train_chair_stage1.py: 0.2)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        # "nccl" = RCCL on ROCm, one rank per GPU; DFX_BENCH_BACKEND=gloo runs several ranks on one GPU (plumbing test: the flat
        # gradient bucket is then staged through the host)
        dist.init_process_group(os.environ.get("DFX_BENCH_BACKEND", "nccl"))
    net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=a.dropout, context_dim=256 + 6,
               n_class=4, class_cond=True, use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True,
               cat_class_to_x=True)
    diff = AnchoredDiffusion(net=net, num_timesteps=a.timesteps, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear',
                             use_beta=False, rescale_timesteps=False, model_mean_type="epsilon", learn_variance=True, loss_type='mse',
                             include_anchors=False, precision=a.precision)
    diff.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(0).items()})
    diff = diff.cuda().train()
    params = list(diff.model.parameters())
    if world > 1:
        parallel.broadcast_params(dict(diff.model.named_parameters()), src=0)
    opt = training.Adam(params, lr=a.lr, max_norm=10.0)
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    B, N = a.batch, a.npoints
    bucket = None
    for it in range(a.iters):
        pc, mean, logvar, valid = synth.make_latents(B, seed=1000 * rank + it)
        seg = synth.make_seg_mask(valid, N)
        var = np.exp(logvar).astype(np.float32)
        idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
        anc, vr = np.take_along_axis(mean, idx, axis=2), np.take_along_axis(var, idx, axis=2)
        cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        x0 = cu((anc + 0.5 * np.sqrt(vr) * rng.standard_normal((B, 3, N))).astype(np.float32))
        t = torch.from_numpy(rng.integers(0, a.timesteps, size=(B,))).cuda()          # samplers/sampler.py:25-40 (Uniform)
        ctx = [cu(pc), cu(np.concatenate([mean, var], 1).astype(np.float32))]
        opt.lr = training.linear_lr(it, start_epoch=4000, end_epoch=8000, start_lr=a.lr, end_lr=1e-4)
        t0 = time.perf_counter()
        opt.zero_grad()
        losses = diff.training_losses(x0, t, anchors=cu(anc), variance=cu(vr), ctx=ctx, anchor_assignment=cu(seg.astype(np.int32)),
                                      valid_id=cu(valid), flags=None)
        losses["mse_loss"].backward()
        bucket = parallel.allreduce_gradients(params, average=True, bucket=bucket)
        norm = opt.step()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"iter {it}: mse_loss {float(losses['mse_loss'].detach()):.4f}  grad norm {float(norm):.3f}  lr {opt.lr:.2e}  "
                  f"{(time.perf_counter() - t0) * 1e3:.1f} ms ({B * world} shapes)", flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host time per stage-1 iteration (enqueue only, no sync) at a small batch, where the GPU is never the bottleneck: distribution over 60 iterations (GPU box)."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from difffacto_amd import synth, training
from difffacto_amd.encoders import PartEncoderForTransformerDecoder
from difffacto_amd.modules import AnchoredDiffusion

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 2048
enc = PartEncoderForTransformerDecoder(encoder=dict(type="PointNetV2", zdim=256, per_part_mlp=True), n_class=4, part_aligner=None, include_z=False,
                                       include_part_code=True, include_params=True, use_gt_params=True, kl_weight=5e-4, use_flow=True, latent_flow_depth=14,
                                       latent_flow_hidden_dim=256, gen=True, prior_var=1.0)
net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=0.0, context_dim=256 + 6, n_class=4, class_cond=True,
           use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True, cat_class_to_x=True)
diff = AnchoredDiffusion(net=net, num_timesteps=1000, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False, rescale_timesteps=False,
                         model_mean_type="epsilon", learn_variance=True, loss_type='mse', include_anchors=False, precision="bf16")
enc, diff = enc.cuda().train(), diff.cuda().train()
opt = training.Adam(list(enc.parameters()) + list(diff.model.parameters()), lr=1e-4, max_norm=10.0)
rng = np.random.Generator(np.random.PCG64(0))
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
_, shift, lv, valid = synth.make_latents(B, seed=0)
seg = synth.make_seg_mask(valid, N)
std = np.exp(0.5 * lv).astype(np.float32)
idx = np.broadcast_to(seg.astype(np.int64)[:, None, :], (B, 3, N))
pts = (np.take_along_axis(shift, idx, 2) + np.take_along_axis(std, idx, 2) * rng.standard_normal((B, 3, N))).astype(np.float32)
pcds = {"input": cu(pts.transpose(0, 2, 1)), "ref": cu(pts.transpose(0, 2, 1)), "present": cu(valid), "dp_present": cu(valid), "ref_seg_mask": cu(seg.astype(np.int64)),
        "ref_attn_map": cu(np.eye(4, dtype=np.float32)[seg]), "part_shift": cu(shift), "part_scale": cu(std), "noise": torch.zeros(B, 32).cuda()}


def it():
    opt.zero_grad()
    losses = training.stage1_losses(enc, diff, pcds)
    sum(v.sum() for k, v in losses.items() if "loss" in k).backward()
    opt.step()


for _ in range(5):
    it()
torch.cuda.synchronize()
for label, prep in (("gc on", lambda: None), ("gc off", gc.disable)):
    prep()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        it()
        ts.append((time.perf_counter() - t0) * 1e3)
        if len(ts) % 8 == 0:
            torch.cuda.synchronize()
    ts = np.array(ts)
    print(f"B={B} {label}: host ms per iteration min {ts.min():.2f} median {np.median(ts):.2f} p90 {np.percentile(ts, 90):.2f} max {ts.max():.2f}")

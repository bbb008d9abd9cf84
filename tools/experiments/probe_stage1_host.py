"""Is the stage-1 training step host-bound?  Per iteration: host time until the last launch is enqueued vs wall time until the GPU is done
(python tools/experiments/probe_stage1_host.py [iters])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from difffacto_amd import synth, training
from difffacto_amd.encoders import PartEncoderForTransformerDecoder
from difffacto_amd.modules import AnchoredDiffusion

iters = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 10
if "--bn" in sys.argv:   # A/B: dfx_debug_bn_fused_stats (0 = separate BatchNorm statistics passes, 1 = default: from the product epilogues)
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_bn_fused_stats(int(sys.argv[sys.argv.index("--bn") + 1]))
QUIET = "--quiet" in sys.argv
B, N = 128, 2048
torch.cuda.set_device(0)
enc = PartEncoderForTransformerDecoder(encoder=dict(type="PointNetV2", zdim=256, per_part_mlp=True), n_class=4, part_aligner=None,
                                       include_z=False, include_part_code=True, include_params=True, use_gt_params=True, kl_weight=5e-4,
                                       use_flow=True, latent_flow_depth=14, latent_flow_hidden_dim=256, gen=True, prior_var=1.0)
net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=0.0, context_dim=256 + 6,
           n_class=4, class_cond=True, use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True, cat_class_to_x=True)
diff = AnchoredDiffusion(net=net, num_timesteps=1000, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False,
                         rescale_timesteps=False, model_mean_type="epsilon", learn_variance=True, loss_type='mse', include_anchors=False,
                         precision="bf16")
enc, diff = enc.cuda().train(), diff.cuda().train()
opt = training.Adam(list(enc.parameters()) + list(diff.model.parameters()), lr=2e-3, max_norm=10.0)
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
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    losses = training.stage1_losses(enc, diff, pcds, epoch=it)
    t1 = time.perf_counter()
    total = sum(v.sum() for k, v in losses.items() if "loss" in k)
    total.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if not QUIET:
      print(f"iter {it}: host forward {1e3 * (t1 - t0):.2f}  backward {1e3 * (t2 - t1):.2f}  step {1e3 * (t3 - t2):.2f}  -> enqueued at {1e3 * (t3 - t0):.2f} ms, GPU done at {1e3 * (t4 - t0):.2f} ms")

# the loop as a trainer runs it: no host synchronisation inside
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for it in range(K):
    opt.zero_grad()
    losses = training.stage1_losses(enc, diff, pcds, epoch=it)
    total = sum(v.sum() for k, v in losses.items() if "loss" in k)
    total.backward()
    opt.step()
torch.cuda.synchronize()
print(f"stage-1 step, {K} iterations back to back: {1e3 * (time.perf_counter() - t0) / K:.2f} ms per iteration")

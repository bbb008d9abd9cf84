"""Run-to-run reproducibility of the whole stage-1 training step (configs/train_chair_stage1.py: PointNetV2 in train mode + prior loss through the flows + the
denoiser with Dropout + every gradient) at the bench size: python tools/soak_stage1_step.py [reps] [B] [N] [dropout p]
torch's and numpy's generators are re-seeded in front of every repetition (the noise and the dropout key / the timesteps are drawn from them); every loss and every parameter gradient of a
repetition is compared with the first one bit by bit.  Exit code 1 on any difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from difffacto_amd import synth, training
from difffacto_amd.encoders import PartEncoderForTransformerDecoder
from difffacto_amd.modules import AnchoredDiffusion
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
enc = PartEncoderForTransformerDecoder(encoder=dict(type="PointNetV2", zdim=256, per_part_mlp=True), n_class=4, part_aligner=None, include_z=False,
                                       include_part_code=True, include_params=True, use_gt_params=True, kl_weight=5e-4, use_flow=True, latent_flow_depth=14,
                                       latent_flow_hidden_dim=256, gen=True, prior_var=1.0)
net = dict(type='TransformerNet', in_channels=3, out_channels=3, n_heads=8, d_head=16, depth=5, dropout=p, context_dim=256 + 6, n_class=4, class_cond=True,
           use_linear=True, cat_params_to_x=True, use_checkpoint=False, single_attn=True, cat_class_to_x=True)
diff = AnchoredDiffusion(net=net, num_timesteps=1000, beta_1=1e-4, beta_T=.02, k=1.0, res=False, mode='linear', use_beta=False, rescale_timesteps=False,
                         model_mean_type="epsilon", learn_variance=True, loss_type='mse', include_anchors=False, precision="bf16")
enc, diff = enc.cuda().train(), diff.cuda().train()
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
params = [(n, q) for n, q in list(enc.named_parameters()) + list(diff.model.named_parameters())]
bn = [(n, b) for n, b in enc.named_buffers() if "running" in n]
bn0 = [b.clone() for _, b in bn]
def step():
    torch.manual_seed(1234), np.random.seed(1234)   # (the timesteps are drawn on the host with numpy, like the reference's sampler)
    for (_, b), b0 in zip(bn, bn0): b.copy_(b0)   # (the running statistics are state: same start every repetition)
    for _, q in params: q.grad = None
    losses = training.stage1_losses(enc, diff, pcds)
    sum(v.sum() for k, v in losses.items() if "loss" in k).backward()
    torch.cuda.synchronize()
    out = [("loss/" + k, v.detach().clone()) for k, v in losses.items()]
    out += [("g/" + n, q.grad.clone()) for n, q in params if q.grad is not None] + [("bn/" + n, b.clone()) for n, b in bn]
    return out
step()         # (warm-up: one-time state — cached tables, lazily created workspaces — is not part of the comparison)
ref = step()
bad, where = 0, {}
for r in range(reps):
    got = step()
    d = [n for (n, x), (_, y) in zip(got, ref) if not torch.equal(x, y)]
    bad += bool(d)
    for n in d: where[n] = where.get(n, 0) + 1
print("losses of the reference repetition:", {n: float(x.sum()) for n, x in ref if n.startswith("loss/")})
print(f"stage-1 step soak B={B} N={N} dropout={p}: {len(ref)} tensors, {reps} repetitions against the first, {bad} different", dict(list(where.items())[:8]))
sys.exit(1 if bad else 0)

"""Gradient oracle for the training-mode denoiser (ORACLE — test infrastructure only, never imported by the product).

PyTorch-CPU autograd over the restated forward of oracle/torch_cpu.py (TransformerNet.forward attention.py:385-440,
dropout = 0) and the masked mse_loss of AnchoredDiffusion.training_losses (anchored_diffusion.py:840-847).  Pinned to
the reference's own autograd by tests/golden/train_grads_*.npz (tests/test_oracle_golden.py).
"""
import numpy as np
import torch

from . import torch_cpu


def masked_mse(target, pred, flags):
    """((target - pred)^2 * flags).mean(1).sum() / flags.sum(); flags (B,1,N) or None -> plain mean."""
    d = (target - pred) ** 2
    if flags is None:
        return d.mean()
    return (d * flags).mean(dim=1).sum() / flags.sum()


def loss_and_grads(W, x_t, t, ctx_code, ctx_mv, anchors_pt, variances_pt, valid, assignment, noise, flags, drops=None):
    """W: dict name -> float32 numpy array.  x_t, noise (B,3,N); anchors_pt, variances_pt (B,N,3); flags (B,1,N) or None.
    drops: explicit dropout factors as numpy arrays, keys as in oracle.torch_cpu.transformer_net_forward, or None.
    Returns dict(loss, eps, grads{name: array}, d_ctx_code, d_ctx_mv) as numpy."""
    Wt = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(True) for k, v in W.items()}
    cc = torch.from_numpy(np.ascontiguousarray(ctx_code)).clone().requires_grad_(True)
    cm = torch.from_numpy(np.ascontiguousarray(ctx_mv)).clone().requires_grad_(True)
    depth = 0
    while f"transformer_blocks.{depth}.norm2.weight" in W:
        depth += 1
    eps = torch_cpu.transformer_net_forward(Wt, torch.from_numpy(x_t), torch.from_numpy(np.asarray(t, dtype=np.int64)), [cc, cm],
                                            torch.from_numpy(anchors_pt), torch.from_numpy(variances_pt),
                                            None if valid is None else torch.from_numpy(valid), torch.from_numpy(np.asarray(assignment)),
                                            depth=depth, drops=None if drops is None else {k: torch.from_numpy(v) for k, v in drops.items()})
    loss = masked_mse(torch.from_numpy(noise), eps, None if flags is None else torch.from_numpy(flags))
    loss.backward()
    return dict(loss=float(loss.detach()), eps=eps.detach().numpy(), grads={k: v.grad.numpy() for k, v in Wt.items() if v.grad is not None},
                d_ctx_code=cc.grad.numpy(), d_ctx_mv=cm.grad.numpy())

"""Stage-1 iteration (bench.py's loop, no per-iteration synchronisation) in three variants: prior-loss branch on the main
stream, on the second stream (default), and stubbed out (the floor an ideal overlap could reach).  python tools/experiments/ab_stage1_overlap.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from difffacto_amd import training

real_stage1, real_prior = training.stage1_losses, training.prior_loss


def run(label, overlap, stub):
    training.stage1_losses = lambda *a, **k: real_stage1(*a, **{**k, "overlap_prior": overlap})
    if stub:   # keep the graph (a zero that depends on part_code / logvar), drop the flows
        training.prior_loss = lambda params, part_code, logvar, valid, **k: ((part_code.sum() + logvar.sum()) * 0.0,
                                                                              torch.zeros(part_code.shape[0], 4, device=part_code.device),
                                                                              torch.zeros(part_code.shape[0], 4, device=part_code.device))
    try:
        r = bench.stage1_iteration(128, 2048, iters=10)
    finally:
        training.stage1_losses, training.prior_loss = real_stage1, real_prior
    print(f"{label:58s} {r['ms']:6.2f} ms per iteration")


run("prior-loss branch on the main stream", False, False)
run("prior-loss branch on a second stream (default)", True, False)
run("prior-loss branch stubbed out (floor)", False, True)

"""Shared helpers of the training-path tests: inputs of a golden case in the layouts the oracle / libdfx take."""
import os

import numpy as np

from difffacto_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(tag="B3_N64_T10"):
    g = dict(np.load(os.path.join(GOLD, f"train_grads_{tag}.npz")))
    W = synth.make_denoiser_weights(int(g["weight_seed"]))
    seg = g["seg"].astype(np.int64)
    mean, var = g["mean"], np.exp(g["logvar"]).astype(np.float32)
    idx = np.broadcast_to(seg[:, None, :], (seg.shape[0], 3, seg.shape[1]))
    anchors = np.take_along_axis(mean, idx, axis=2)             # (B,3,N)  gather_all, part_encoders.py:417-428
    variance = np.take_along_axis(var, idx, axis=2)
    case = dict(W=W, x_t=g["x_t"], t=g["t"], ctx_code=g["part_code"], ctx_mv=np.concatenate([mean, var], axis=1).astype(np.float32),
                anchors_pt=np.ascontiguousarray(anchors.transpose(0, 2, 1)), variances_pt=np.ascontiguousarray(variance.transpose(0, 2, 1)),
                valid=g["valid"], assignment=g["seg"].astype(np.int32), noise=g["noise"], flags=g["flags"])
    return g, case


def check_against_golden(g, grads, rtol, atol):
    """grads: dict name -> array (any shape).  Compares with the golden's full tensors / samples + (sum, L2 norm)."""
    worst = 0.0
    n = 0
    for key in g:
        if key.startswith("g/"):
            name = key[2:]
            got = np.asarray(grads[name], dtype=np.float64).ravel()
            ref = g[key].astype(np.float64)
            scale = max(np.abs(ref).max(), 1e-30)
            err = np.abs(got - ref).max()
            assert err <= atol + rtol * scale, (name, err, scale)
            worst = max(worst, err / scale)
            n += 1
        elif key.startswith("gs/"):
            name = key[3:]
            full = np.asarray(grads[name], dtype=np.float64).ravel()
            ref = g[key].astype(np.float64)
            got = full[g["gi/" + name]]
            scale = max(np.abs(ref).max(), 1e-30)
            err = np.abs(got - ref).max()
            assert err <= atol + rtol * scale, (name, err, scale)
            s, l2 = g["gn/" + name]
            assert abs(np.sqrt((full ** 2).sum()) - l2) <= rtol * l2 + atol, (name, "L2", np.sqrt((full ** 2).sum()), l2)
            assert abs(full.sum() - s) <= rtol * np.abs(full).sum() + atol, (name, "sum", full.sum(), s)
            worst = max(worst, err / scale)
            n += 1
    return n, worst

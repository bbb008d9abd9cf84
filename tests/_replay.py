"""Replay of the reference's recorded random draws (tests/golden/make_golden_forward.py) at the same sites of the mirror:
``torch.randn`` / ``torch.randn_like`` pop the next recorded array and check its shape."""
import contextlib

import numpy as np
import torch


@contextlib.contextmanager
def replay_draws(draws):
    queue = [np.asarray(a) for a in draws]
    real = (torch.randn, torch.randn_like)

    def randn(*shape, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        a = queue.pop(0)
        assert tuple(shape) == a.shape, (tuple(shape), a.shape)
        return torch.from_numpy(a.copy()).to(device)

    def randn_like(x, **kw):
        a = queue.pop(0)
        assert tuple(x.shape) == a.shape, (tuple(x.shape), a.shape)
        return torch.from_numpy(a.copy()).to(x.device)

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield queue
    finally:
        torch.randn, torch.randn_like = real


def load_forward_fixture(path):
    """-> (batch dict of torch tensors, list of draws, expected dict key -> array, meta)."""
    g = np.load(path)
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in/")}
    draws = [g[f"draw_{i}"] for i in range(int(g["n_draws"]))]
    expect = {k[4:]: g[k] for k in g.files if k.startswith("out/")}
    meta = {k: g[k] for k in g.files if not (k.startswith("in/") or k.startswith("out/") or k.startswith("draw_"))}
    return batch, draws, expect, meta

"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: weight broadcast + ragged cloud gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from difffacto_amd import parallel, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = synth.denoiser_param_shapes(depth=2)
        if rank == 0:
            W = synth.make_denoiser_weights(seed=3, depth=2)
            params = {k: torch.from_numpy(W[k]) for k, _ in shapes}
        else:
            params = {k: torch.full(s, float("nan")) for k, s in shapes}
        parallel.broadcast_params(params, src=0)
        ref = synth.make_denoiser_weights(seed=3, depth=2)
        ok_b = all(torch.equal(params[k], torch.from_numpy(ref[k])) for k, _ in shapes)
        total = 7                                     # ragged: 4 + 3 shapes
        lo, hi = parallel.shard_range(total, rank, world)
        mine = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 5, 3).contiguous()
        got = parallel.gather_clouds(mine, dst=0)
        if rank == 0:
            exp = torch.arange(total, dtype=torch.float32).view(-1, 1, 1).expand(-1, 5, 3)
            ok_g = got is not None and torch.equal(got, exp)
        else:
            ok_g = got is None
        q.put((rank, ok_b, ok_g))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_shard_range_partitions():
    for total in (0, 1, 7, 128, 1024):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1

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
        # the same with the per-rank counts handed in (what bench.py does: block partition, no size exchange)
        got2 = parallel.gather_clouds(mine, dst=0, sizes=[b - a for a, b in (parallel.shard_range(total, r, world) for r in range(world))])
        if rank == 0:
            exp = torch.arange(total, dtype=torch.float32).view(-1, 1, 1).expand(-1, 5, 3)
            ok_g = got is not None and torch.equal(got, exp) and torch.equal(got2, exp)
        else:
            ok_g = got is None and got2 is None
        # leaf nn.Parameters (requires_grad) are unpacked in place under no_grad (examples/train_denoiser.py's call)
        lin = torch.nn.Linear(4, 3)
        with torch.no_grad():
            for p in lin.parameters():
                p.fill_(float(rank + 1))
        parallel.broadcast_params(dict(lin.named_parameters()), src=0)
        ok_b = ok_b and all(bool((p == 1.0).all()) and p.requires_grad for p in lin.parameters())
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


def _world_worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DFX_GATHER"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 2 * world + 2                          # ragged for world > 2
        sizes = [b - a for a, b in (parallel.shard_range(total, r, world) for r in range(world))]
        lo, hi = parallel.shard_range(total, rank, world)
        mine = (torch.arange(lo, hi, dtype=torch.float32) + 0.5).view(-1, 1, 1).expand(-1, 6, 3).contiguous()
        got = parallel.gather_clouds(mine, dst=0, sizes=sizes)
        exp = (torch.arange(total, dtype=torch.float32) + 0.5).view(-1, 1, 1).expand(-1, 6, 3)
        ok = torch.equal(got, exp) if rank == 0 else got is None
        seen = parallel.describe_world("cpu")
        ok = ok and seen == {"backend": "gloo", "world_size": world, "devices": [-1] * world}
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(4, "gather"), (4, "all_gather"), (8, "all_gather"), (8, "gather")])
def test_gather_modes_and_world_description_beyond_two_ranks(world, mode):
    """world sizes above 2 (only 2 had ever run), the rooted gather and its DFX_GATHER=all_gather fallback on a
    ragged block partition, and describe_world() — the block bench.py puts into the N > 1 JSON line."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_gather_mode_rejects_unknown_value(monkeypatch):
    monkeypatch.setenv("DFX_GATHER", "scatter")
    with pytest.raises(ValueError):
        parallel._gather_mode()


def test_shard_range_partitions():
    for total in (0, 1, 7, 128, 1024):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = [(5, 3), (7,), (2, 2, 2)]
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        for i, p in enumerate(params):
            if not (rank == 1 and i == 1):                      # rank 1 has no gradient for tensor 1: counts as zeros
                p.grad = torch.full(p.shape, float(rank + 1) * (i + 1))
        bucket = parallel.allreduce_gradients(params, average=True)
        exp = [torch.full(shapes[0], (1 + 2) * 1 / 2), torch.full(shapes[1], (1 * 2 + 0) / 2), torch.full(shapes[2], (1 + 2) * 3 / 2)]
        ok = all(torch.equal(p.grad, e) for p, e in zip(params, exp)) and bucket.numel() == 15 + 7 + 8
        bucket2 = parallel.allreduce_gradients(params, average=False, bucket=bucket)   # reuse, sum
        ok = ok and bucket2 is bucket and torch.equal(params[0].grad, exp[0] * 2)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    """The training path's collective: one flat bucket, summed / averaged across two ranks (gloo here, RCCL on GPUs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_linear_lr_matches_the_reference_schedule():
    """training.linear_lr against torch's LambdaLR driven by the reference's lr_func (optimizers/schedulers.py:8-19),
    with train_chair_stage1.py's values (2e-3 -> 1e-4 between epochs 4000 and 8000)."""
    from difffacto_amd.training import linear_lr
    start_epoch, end_epoch, start_lr, end_lr = 4000, 8000, 2e-3, 1e-4

    def lr_func(epoch):
        if epoch <= start_epoch:
            return 1.0
        if epoch <= end_epoch:
            frac = (epoch - start_epoch) / (end_epoch - start_epoch)
            return (1 - frac) * 1.0 + frac * (end_lr / start_lr)
        return end_lr / start_lr

    for e in (0, 1, 3999, 4000, 4001, 6000, 7999, 8000, 8001, 20000):
        assert abs(linear_lr(e, start_epoch, end_epoch, start_lr, end_lr) - start_lr * lr_func(e)) < 1e-15
    assert linear_lr(6000, start_epoch, end_epoch, start_lr, end_lr) == pytest.approx(1.05e-3)


def _probe_worker(rank, world, port, q, break_gather):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DFX_GATHER", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if break_gather and rank == 1:
            # a collective library whose rooted gather raises on ONE rank only: every rank must still switch, together
            real = dist.gather

            def broken(*a, **k):
                real(*a, **k)                      # take part in the collective (the others would hang otherwise), then fail
                raise RuntimeError("rooted gather is broken here")
            dist.gather = broken
        info = parallel.probe_gather("cpu")
        dist.gather = getattr(dist, "gather")
        lo, hi = parallel.shard_range(5, rank, world)
        mine = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 3).contiguous()
        if break_gather:
            def never(*a, **k):
                raise AssertionError("the rooted gather must not be used after the fallback")
            dist.gather = never
        got = parallel.gather_clouds(mine, dst=0, sizes=[3, 2])
        ok = torch.equal(got, torch.arange(5, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 3)) if rank == 0 else got is None
        times = parallel.all_gather_floats([10.0 + rank, 2.0 * rank], "cpu")
        ok = ok and times == [[10.0, 0.0], [11.0, 2.0]]
        q.put((rank, bool(ok), info["gather"], info["fallback"] is not None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("break_gather", [False, True])
def test_probe_gather_falls_back_on_every_rank_together(break_gather):
    """bench.py --gpus N probes the rooted gather once; if it raises on any rank, all ranks use all_gather (and the
    JSON line says so); per-rank kernel times travel through one small all_gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q, break_gather)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mode = "all_gather" if break_gather else "gather"
    assert res == [(0, True, mode, break_gather), (1, True, mode, break_gather)]
    assert parallel.collective_library() is None or isinstance(parallel.collective_library(), str)

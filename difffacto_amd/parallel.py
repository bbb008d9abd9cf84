"""Multi-GPU plumbing for sampling: one process per GPU, independent shapes sharded over ranks.

The path shards naturally: shapes are independent, so the ONLY communication is
  (1) one broadcast of the frozen parameter blob from rank 0 (RCCL over xGMI; flat 1->N), and
  (2) one gather of the generated clouds (B_r, N, 3) to rank 0 at the end.
Nothing per diffusion step.  The reference's own multi-GPU mode is nn.DataParallel / DDP with NCCL
(python/difffacto/runner/runner.py:61-73, utils/dist_utils.py:9-62) and has no sampling collective.
Works with the "nccl" (= RCCL) backend on GPUs and with "gloo" on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block partition of `total` shapes: rank r gets [lo, hi).  Sizes differ by <= 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _needs_cpu_staging(t):
    """gloo moves host memory: device tensors are staged through the host (test / fallback configurations only;
    the production backend is "nccl" = RCCL, which takes device pointers directly)."""
    return dist.get_backend() == "gloo" and t.is_cuda


@torch.no_grad()
def broadcast_params(params, src=0):
    """One collective for the whole frozen parameter set: pack -> broadcast -> unpack in place.
    `params` is an ordered dict name -> tensor, same shapes on every rank (plain tensors or leaf nn.Parameters: the
    in-place unpack runs under no_grad)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return params
    names = list(params)
    flat = torch.cat([params[k].reshape(-1).to(torch.float32) for k in names])
    if _needs_cpu_staging(flat):
        host = flat.cpu()
        dist.broadcast(host, src=src)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    for k in names:
        n = params[k].numel()
        params[k].copy_(flat[off:off + n].view_as(params[k]))
        off += n
    return params


def gather_clouds(pred, dst=0, sizes=None):
    """Gather per-rank generated clouds (B_r, N, 3) to `dst`; returns the concatenation on dst
    (rank order) and None elsewhere.  Ragged B_r is allowed: `sizes` = the per-rank shape counts when the caller knows them
    (a block partition does: shard_range) — otherwise they are exchanged first with one small all_gather and a host sync."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return pred
    world, rank = dist.get_world_size(), dist.get_rank()
    out_device = pred.device
    if _needs_cpu_staging(pred):
        pred = pred.cpu()
    if sizes is None:
        sizes = [torch.zeros(1, dtype=torch.int64, device=pred.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([pred.shape[0]], dtype=torch.int64, device=pred.device))
        sizes = [int(s.item()) for s in sizes]
    else:
        sizes = [int(x) for x in sizes]
        assert len(sizes) == world and sizes[rank] == pred.shape[0], (sizes, rank, tuple(pred.shape))
    bmax = max(sizes)
    pad = pred
    if pred.shape[0] < bmax:
        pad = torch.cat([pred, pred.new_zeros((bmax - pred.shape[0],) + tuple(pred.shape[1:]))])
    pad = pad.contiguous()
    if _gather_mode() == "all_gather":
        # fallback (DFX_GATHER=all_gather): every rank receives every block (N x the bytes of the rooted gather — 3 MB per 128 shapes,
        # irrelevant next to a T-step chain); for a collective library whose rooted gather misbehaves
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)]).to(out_device) if rank == dst else None
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=bufs, dst=dst)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)]).to(out_device)
    dist.gather(pad, gather_list=None, dst=dst)
    return None


_MODE_OVERRIDE = None   # set by probe_gather() when the rooted gather failed on some rank


def _gather_mode():
    import os
    if _MODE_OVERRIDE is not None:
        return _MODE_OVERRIDE
    mode = os.environ.get("DFX_GATHER", "gather")
    if mode not in ("gather", "all_gather"):
        raise ValueError(f"DFX_GATHER={mode!r}: expected 'gather' or 'all_gather'")
    return mode


def probe_gather(device=None):
    """Day-one guard for a collective library this code has not met yet: every rank tries ONE tiny rooted gather; if it raises on
    any rank (agreed through an all_reduce(MIN) of the success flags, so that all ranks switch together), gather_clouds() uses
    all_gather from then on.  Returns {"gather": mode in use, "fallback": None or the first error text}.  A collective that HANGS
    cannot be rescued from inside the process — that includes a gather that raises on ONE rank only: the others wait inside it until the
    process group's timeout (init_process_group(timeout=...)) before they reach the agreement all_reduce;
    DFX_GATHER=all_gather skips the rooted gather altogether."""
    global _MODE_OVERRIDE
    if not dist.is_initialized() or dist.get_world_size() == 1 or _gather_mode() == "all_gather":
        return {"gather": _gather_mode(), "fallback": None}
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    err = None
    try:
        t = torch.full((4,), float(rank), device=dev)
        if rank == 0:
            bufs = [torch.empty_like(t) for _ in range(world)]
            dist.gather(t, gather_list=bufs, dst=0)
            if dev != "cpu":
                torch.cuda.synchronize()
            if [float(b[0]) for b in bufs] != [float(r) for r in range(world)]:
                raise RuntimeError("rooted gather returned wrong blocks")
        else:
            dist.gather(t, gather_list=None, dst=0)
            if dev != "cpu":
                torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001 - whatever the backend raises
        err = repr(e)[:200]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        _MODE_OVERRIDE = "all_gather"
        return {"gather": "all_gather", "fallback": err or "the rooted gather failed on another rank"}
    return {"gather": "gather", "fallback": None}


def all_gather_floats(values, device=None):
    """Every rank's list of floats -> (world, len) nested list on every rank (one small all_gather): per-rank kernel times for the
    bench line, so that a straggler GPU is visible."""
    v = [float(x) for x in values]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [v]
    dev = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    mine = torch.tensor(v, dtype=torch.float64, device=dev)
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return [[float(x) for x in g.cpu()] for g in got]


def collective_library():
    """Version of the collective library behind the "nccl" backend (RCCL on ROCm) as torch reports it, or None."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:   # noqa: BLE001 - CPU-only torch builds, missing symbol
        return None


def describe_world(device=None):
    """What the process group actually looks like, for the bench line of an N > 1 run: backend, world size as torch.distributed sees
    it, and every rank's device index (one small all_gather).  Proof that the collective library saw N ranks on N devices."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "devices": [torch.cuda.current_device() if torch.cuda.is_available() else -1]}
    world = dist.get_world_size()
    on_gpu = device is not None and torch.device(device).type == "cuda"
    mine = torch.tensor([torch.cuda.current_device() if on_gpu else -1], dtype=torch.int64,
                        device=device if (on_gpu and dist.get_backend() != "gloo") else "cpu")
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return {"backend": dist.get_backend(), "world_size": world, "devices": [int(g.item()) for g in got]}


def allreduce_gradients(params, average=True, bucket=None):
    """Data-parallel training step exchange (the one real collective of the training path): the gradients of all `params`
    are packed into ONE flat fp32 bucket (10.5 MB for the denoiser: a single ring all-reduce over xGMI, per-link bound,
    instead of 77 small ones), summed across ranks with `all_reduce` (RCCL on GPUs, gloo on CPU) and scattered back.
    Parameters without a gradient contribute zeros, so every rank reduces the same layout.  Returns the bucket for reuse.
    No-op when torch.distributed is not initialised or the world has one rank."""
    import torch
    import torch.distributed as dist
    params = [p for p in params]
    if not params or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return bucket
    n = sum(p.numel() for p in params)
    dev = params[0].device
    if bucket is None or bucket.numel() != n or bucket.device != dev:
        bucket = torch.empty(n, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        k = p.numel()
        if p.grad is None:
            bucket[off:off + k].zero_()
        else:
            bucket[off:off + k].copy_(p.grad.reshape(-1))
        off += k
    if _needs_cpu_staging(bucket):   # gloo moves host memory (test configurations; RCCL takes the device bucket as it is)
        host = bucket.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        bucket.copy_(host)
    else:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    if average:
        bucket.div_(dist.get_world_size())
    off = 0
    for p in params:
        k = p.numel()
        g = bucket[off:off + k].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += k
    return bucket

"""Generation pipeline of the `gen_*` path (AnchorDiffAE.forward's generation branch, anchor_gen.py:1034-1084) with the host out of the way:

    pipe = SamplingPipeline(engine, sampler, B, N, valid_id)
    for pred in pipe.run(n_batches, seed0):      # one (B, N, 3) cloud batch per iteration
        ...

Per batch the reference runs `sample_latents` (two `torch.randn` draws, 4 x 14 coupling layers in reverse, the part aligner: ~100 small
launches here) and then the T-step `decode`.  Two things keep that front end off the critical path:

* **one hipGraph for the front end** — the draws, `dfx_sample_latents` and `dfx_shape_ctx_prepare` are captured once per (B, N)
  (`torch.cuda.CUDAGraph`: HIP stream capture of libdfx's launches on torch's stream; outputs live in the graph's private pool) and
  replayed with ONE host call per batch instead of ~100 (the launches are 2-10 us of GPU time each, and ~8 us of host time each when
  issued one by one: at B = 1 the front end was 0.8 ms of a 33 ms pass);
* **a second HIP stream** — the front end of batch i + 1 is replayed on a side stream while the chain kernel of batch i runs (two graph
  instances alternate, so batch i's operands are never overwritten under the chain that reads them); the chain launch waits on an event,
  never on the host.

The chain itself stays a plain `dfx_sample_chain` launch: its Philox seed and shape offset are by-value kernel arguments that change with
every batch.  Results are the same bits as the unpipelined calls with the same draws (tested).
"""
import torch

from . import _ffi


class _FrontEnd:
    """One captured instance of [draws -> sample_latents -> shape context] with static outputs."""

    def __init__(self, engine, sampler, B, N, valid, K, use_graph, stream):
        self.engine, self.sampler, self.B, self.N, self.K, self.valid = engine, sampler, B, N, K, valid
        self.stream = stream
        self.done = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.consumed.record(torch.cuda.current_stream(engine.device))
        self.graph = None
        self.lat = self.ctx = None
        self.w = torch.empty(B, sampler.zdim, sampler.n_class, device=engine.device)
        self.an = torch.empty(B * K, sampler.noise_dim, device=engine.device)
        if use_graph:
            cap = torch.cuda.Stream(device=engine.device)   # capture needs a non-default stream; the replay runs wherever it is enqueued
            cap.wait_stream(torch.cuda.current_stream(engine.device))
            with torch.cuda.stream(cap):
                self._body()                      # warm-up: libdfx sizes its workspace on the first call of a shape (hipMalloc is not capturable)
            cap.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                self._body()
            self.graph = g

    def _body(self):
        self.w.normal_()                          # part_encoders.py:1054 (default CUDA generator: graph-safe Philox offsets)
        self.an.normal_()                         # :1065
        self.lat = self.sampler.sample_latents(self.w, self.an, self.valid, K=self.K, npoints=self.N)
        p = self.lat["params"]
        self.ctx = self.engine.prepare_shapes(self.lat["part_code"], p[:, :3], p[:, 3:], self.lat["valid_id"])

    def launch(self):
        """Enqueue this instance's front end on the side stream (after the chain that last read its outputs)."""
        self.stream.wait_event(self.consumed)
        with torch.cuda.stream(self.stream):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._body()
            self.done.record(self.stream)


class SamplingPipeline:
    def __init__(self, engine, sampler, B, N, valid_id, K=1, use_graph=True, overlap=True):
        self.engine, self.sampler = engine, sampler
        self.B, self.N, self.K = int(B), int(N), int(K)
        dev = engine.device
        valid = valid_id.detach().to(device=dev, dtype=torch.float32).contiguous()
        assert tuple(valid.shape) == (self.B, sampler.n_class)
        self.overlap = overlap
        self.side = torch.cuda.Stream(device=dev) if overlap else torch.cuda.current_stream(dev)
        n_inst = 2 if overlap else 1
        self.inst = [_FrontEnd(engine, sampler, self.B, self.N, valid, self.K, use_graph, self.side) for _ in range(n_inst)]
        self.last_chain_events = None

    def run(self, n_batches, seed0=0, shape_offset=0, time_chain=False):
        """Generator over `n_batches` cloud batches (R, N, 3), R = B * K.  `time_chain`: record HIP events around each chain launch
        (``self.last_chain_events`` = list of (start, end))."""
        main = torch.cuda.current_stream(self.engine.device)
        evs = []
        if n_batches <= 0:
            return
        self.inst[0].launch()
        for i in range(n_batches):
            cur = self.inst[i % len(self.inst)]
            if self.overlap and i + 1 < n_batches:
                self.inst[(i + 1) % len(self.inst)].launch()       # next batch's front end rides beside this batch's chain
            main.wait_event(cur.done)
            if time_chain:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main)
            pred, _ = self.engine.sample_chain(cur.ctx, cur.lat["seg_mask"], seed=seed0 + i, shape_offset=shape_offset)
            if time_chain:
                b.record(main)
                evs.append((a, b))
            cur.consumed.record(main)
            if not self.overlap and i + 1 < n_batches:
                self.inst[0].launch()
            self.last_chain_events = evs
            yield pred

    def latents(self, k=0):
        """The latents dict of front-end instance k (valid after its `done` event; for tests)."""
        return self.inst[k].lat

"""Training-mode denoiser on the HIP path: autograd functions over libdfx's forward / backward kernels,
and the reference's optimiser step.

Mirrors what ``loss.backward()`` + ``clip_grad_norm_`` + ``Adam.step`` do to ``TransformerNet`` in the reference
(python/difffacto/models/networks/attention.py:385-440 forward; python/difffacto/runner/runner.py:312-316 step;
anchored_diffusion.py:840-847 loss), dropout included.  PyTorch provides the tensors, the autograd graph and the
stream; every number is produced by the kernels of ``csrc/train_kernels.hip``.
"""
import ctypes

import torch

from . import _ffi
from .engine import _BLOCK_FIELDS, _TOP_FIELDS, EXPECTED_SHAPES, PRECISIONS

__all__ = ["dropout_factors", "stage1_losses", "prior_loss", "PriorLossFn", "pointnet_v2_train_forward", "PointNetV2TrainFn", "param_names", "DenoiserTrainFn", "denoiser_train_forward", "MaskedMSEFn", "masked_mse", "Adam", "linear_lr"]


def param_names(depth):
    """state_dict keys of TransformerNet in the order the autograd function takes them."""
    names = list(_TOP_FIELDS.values())
    for b in range(depth):
        names += [f"transformer_blocks.{b}.{k}" for k in _BLOCK_FIELDS.values()]
    return names


def _struct(tensors, depth):
    """dfx_denoiser_weights over `tensors` (dict name -> contiguous fp32 cuda tensor)."""
    w = _ffi.DenoiserWeights()
    w.depth = depth
    for field, key in _TOP_FIELDS.items():
        setattr(w, field, tensors[key].data_ptr())
    for b in range(depth):
        for field, key in _BLOCK_FIELDS.items():
            setattr(w.blk[b], field, tensors[f"transformer_blocks.{b}.{key}"].data_ptr())
    return w


def _need(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: contiguous fp32 tensor on the GPU required (got {t.dtype}, {t.device})")
    return t


def _check_first_backward(ctx, who):
    if ctx.ws_ptr is None:
        raise RuntimeError(f"{who}.backward ran twice on one forward (retain_graph / two losses sharing the forward): the saved-"
                           "activation workspace is released by the first backward; run the forward again")


class DenoiserTrainFn(torch.autograd.Function):
    """eps = TransformerNet(x, t, [ctx_code, ctx_mv], anchors, variances, valid_id, anchor_assignment), differentiable in
    the parameters, in the two context tensors and — when they require it — in x and in the per-point variances (stage 2:
    the reference's `variance` reaches training_losses UNdetached, anchor_gen.py:1002-1020, and enters through q_sample's
    x_t and through the feature columns); anchors are data (detach_anchor, :1011-1012)."""

    @staticmethod
    def forward(ctx, depth, precision, dropout, x, t, ctx_code, ctx_mv, anchors, variances, valid, assignment, *params):
        names = param_names(depth)
        if precision not in PRECISIONS:
            raise ValueError(f"precision {precision!r}: one of {sorted(PRECISIONS)}")
        prec = PRECISIONS[precision]
        drop_p, drop_seed = (0.0, 0) if not dropout else (float(dropout[0]), int(dropout[1]) & (2 ** 64 - 1))
        if not 0.0 <= drop_p < 1.0:
            raise ValueError(f"dropout probability {drop_p}")
        if len(params) != len(names):
            raise ValueError(f"expected {len(names)} parameter tensors, got {len(params)}")
        B, _, N = x.shape
        tensors = {}
        for n, p in zip(names, params):
            base = n.split(".", 2)[2] if n.startswith("transformer_blocks.") else n
            if base in EXPECTED_SHAPES and tuple(p.shape) != EXPECTED_SHAPES[base]:
                raise ValueError(f"{n}: shape {tuple(p.shape)} != {EXPECTED_SHAPES[base]}")
            tensors[n] = _need(p.detach(), n)
        x = _need(x.detach().contiguous(), "x")
        ctx_code = _need(ctx_code.detach().contiguous(), "ctx_code")
        ctx_mv = _need(ctx_mv.detach().contiguous(), "ctx_mv")
        anchors = _need(anchors.detach().contiguous(), "anchors")
        variances = _need(variances.detach().contiguous(), "variances")
        t32 = t.to(device=x.device, dtype=torch.int32).contiguous()
        asg = assignment.to(device=x.device, dtype=torch.int32).contiguous()
        vld = None if valid is None else valid.to(device=x.device, dtype=torch.float32).contiguous()
        if ctx_code.shape != (B, 256, 4) or ctx_mv.shape != (B, 6, 4) or anchors.shape != (B, N, 3) or variances.shape != (B, N, 3):
            raise ValueError("ctx_code (B,256,4), ctx_mv (B,6,4), anchors / variances (B,N,3) expected")
        lib = _ffi.lib()
        nbytes = lib.dfx_denoiser_train_workspace_bytes(B, N, depth)
        if nbytes == 0:
            raise ValueError(f"unsupported training shape B={B} N={N} depth={depth}")
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        eps = torch.empty(B, 3, N, dtype=torch.float32, device=x.device)
        w = _struct(tensors, depth)
        with torch.cuda.device(x.device):
            _ffi.check(lib.dfx_denoiser_train_forward(ctypes.byref(w), ws_ptr, nbytes, x.data_ptr(), t32.data_ptr(),
                                                      ctx_code.data_ptr(), ctx_mv.data_ptr(), anchors.data_ptr(),
                                                      variances.data_ptr(), None if vld is None else vld.data_ptr(),
                                                      asg.data_ptr(), eps.data_ptr(), B, N, prec, drop_p, drop_seed, _ffi.current_stream()),
                       "dfx_denoiser_train_forward")
        ctx.depth, ctx.shape, ctx.ws, ctx.ws_ptr, ctx.nbytes, ctx.prec = depth, (B, N), ws, ws_ptr, nbytes, prec
        ctx.drop = (drop_p, drop_seed)
        ctx.tensors = tensors
        ctx.leaves = [p if (p.is_leaf and p.requires_grad) else None for p in params]
        ctx.need_ctx = (ctx.needs_input_grad[5], ctx.needs_input_grad[6])
        ctx.need_in = (ctx.needs_input_grad[3], ctx.needs_input_grad[8])   # x, variances
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        B, N = ctx.shape
        depth = ctx.depth
        names = param_names(depth)
        d_eps = _need(d_eps.contiguous(), "d_eps")
        dev = d_eps.device
        _check_first_backward(ctx, "DenoiserTrainFn")
        # one flat buffer, handed out as views in parameter order: clip + Adam can then run as ONE launch each over the
        # whole parameter set (training.Adam), and a data-parallel all-reduce needs no packing copy
        # (every slice starts on a 256-byte boundary: the product kernels want 16-byte aligned weights once the optimiser
        # has re-pointed the parameters into a buffer of the same layout; the gaps stay zero)
        sizes = [ctx.tensors[n].numel() for n in names]
        starts, off = [], 0
        for k in sizes:
            starts.append(off)
            off += (k + 63) // 64 * 64
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        grads = {n: flat[o:o + k].view(ctx.tensors[n].shape) for n, o, k in zip(names, starts, sizes)}
        d_code = torch.empty(B, 256, 4, dtype=torch.float32, device=dev) if ctx.need_ctx[0] else None
        d_mv = torch.empty(B, 6, 4, dtype=torch.float32, device=dev) if ctx.need_ctx[1] else None
        d_x = torch.empty(B, 3, N, dtype=torch.float32, device=dev) if ctx.need_in[0] else None
        d_var = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if ctx.need_in[1] else None
        w, g = _struct(ctx.tensors, depth), _struct(grads, depth)
        with torch.cuda.device(dev):
            _ffi.check(_ffi.lib().dfx_denoiser_train_backward(ctypes.byref(w), ctx.ws_ptr, ctx.nbytes, d_eps.data_ptr(),
                                                              ctypes.byref(g), None if d_code is None else d_code.data_ptr(),
                                                              None if d_mv is None else d_mv.data_ptr(), _ffi.ptr(d_x), _ffi.ptr(d_var), B, N,
                                                              ctx.prec, ctx.drop[0], ctx.drop[1], _ffi.current_stream()),
                       "dfx_denoiser_train_backward")
        ctx.ws = None
        ctx.ws_ptr = None   # the workspace goes back to the allocator: a second backward must not reuse its address
        # Leaf parameters get their slice of the flat buffer assigned to .grad directly (accumulating if one is already
        # there): returned through autograd, AccumulateGrad would clone every tensor out of the flat buffer, because the
        # Python wrappers of the returned views still hold references when it runs.
        out = []
        for n, leaf in zip(names, ctx.leaves):
            if leaf is None:
                out.append(grads[n])
            else:
                if leaf.grad is None:
                    leaf.grad = grads[n]
                else:
                    leaf.grad.add_(grads[n])
                out.append(None)
        ctx.leaves = None
        return (None, None, None, d_x, None, d_code, d_mv, None, d_var, None, None) + tuple(out)


def denoiser_train_forward(params, x, t, ctx_code, ctx_mv, anchors, variances, valid, assignment, precision="f32", dropout=None):
    """`params`: dict state_dict-key -> fp32 cuda tensor (requires_grad as the caller wishes) of a TransformerNet.
    precision "f32": exact fp32 (parity gate); "bf16": bf16 operands / fp32 accumulate for the matrix products.
    dropout: None, or (p, seed): nn.Dropout(p) of train() mode behind every to_out and GEGLU, Philox factors keyed by
    `seed` (a fresh integer per step; the backward of this call regenerates the same factors)."""
    depth = 0
    while f"transformer_blocks.{depth}.norm2.weight" in params:
        depth += 1
    N = x.shape[-1]
    pad = (-N) % 32
    if pad:   # the kernels take N % 32 == 0: run padded (points are independent), cut the padding off; its gradient is zero
        pf = torch.nn.functional.pad
        assignment = torch.cat([assignment, assignment[:, :1].expand(-1, pad)], dim=1).contiguous()
        return denoiser_train_forward(params, pf(x, (0, pad)), t, ctx_code, ctx_mv, pf(anchors, (0, 0, 0, pad)),
                                      pf(variances, (0, 0, 0, pad), value=1.0), valid, assignment, precision, dropout)[..., :N]
    return DenoiserTrainFn.apply(depth, precision, dropout, x, t, ctx_code, ctx_mv, anchors, variances, valid, assignment,
                                 *[params[n] for n in param_names(depth)])


def _flat_slices(shapes, device):
    """One zero-initialised flat fp32 buffer and a view per shape, every view starting on a 256-byte boundary (see
    DenoiserTrainFn.backward)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    starts, off = [], 0
    for k in sizes:
        starts.append(off)
        off += (k + 63) // 64 * 64
    flat = torch.zeros(off, dtype=torch.float32, device=device)
    return [flat[o:o + k].view(s) for o, k, s in zip(starts, sizes, shapes)]


def _assign_or_return(leaves, grads):
    """Leaf parameters get `.grad` assigned (accumulated) directly and None is returned for them; others get the gradient back."""
    out = []
    for leaf, g in zip(leaves, grads):
        if leaf is None:
            out.append(g)
        else:
            if leaf.grad is None:
                leaf.grad = g
            else:
                leaf.grad.add_(g)
            out.append(None)
    return out


PNV2_PARAMS = [f"conv{i}.{k}" for i in (1, 2, 3, 4) for k in ("weight", "bias")] + [f"bn{i}.{k}" for i in (1, 2, 3, 4) for k in ("weight", "bias")] + \
              [f"{h}.{i}.{k}" for h in ("mlp_m", "mlp_v") for i in (0, 1, 3, 4, 6) for k in ("weight", "bias")]
PNV2_BUFFERS = [f"bn{i}.{k}" for i in (1, 2, 3, 4) for k in ("running_mean", "running_var")] + \
               [f"{h}.{i}.{k}" for h in ("mlp_m", "mlp_v") for i in (1, 4) for k in ("running_mean", "running_var")]


def _pnv2_struct(t, num_anchors, zdim, reweight, eps):
    """dfx_pointnet_v2_weights over the dict `t` (state_dict names -> contiguous fp32 cuda tensors; missing names stay NULL)."""
    w = _ffi.PointNetV2Weights()
    w.num_anchors, w.zdim, w.reweight_by_anchor, w.bn_eps = num_anchors, zdim, int(reweight), float(eps)
    ptr = lambda n: t[n].data_ptr() if n in t else None
    for i in range(4):
        w.conv_w[i], w.conv_b[i] = ptr(f"conv{i + 1}.weight"), ptr(f"conv{i + 1}.bias")
        w.bn_w[i], w.bn_b[i] = ptr(f"bn{i + 1}.weight"), ptr(f"bn{i + 1}.bias")
        w.bn_mean[i], w.bn_var[i] = ptr(f"bn{i + 1}.running_mean"), ptr(f"bn{i + 1}.running_var")
    for k, h in enumerate(("mlp_m", "mlp_v")):
        for l, ci in enumerate((0, 3, 6)):
            w.head_w[k][l], w.head_b[k][l] = ptr(f"{h}.{ci}.weight"), ptr(f"{h}.{ci}.bias")
        for l, bi in enumerate((1, 4)):
            w.head_bn_w[k][l], w.head_bn_b[k][l] = ptr(f"{h}.{bi}.weight"), ptr(f"{h}.{bi}.bias")
            w.head_bn_mean[k][l], w.head_bn_var[k][l] = ptr(f"{h}.{bi}.running_mean"), ptr(f"{h}.{bi}.running_var")
    return w


class PointNetV2TrainFn(torch.autograd.Function):
    """(m, v) = PointNetV2(x, attn) in train mode (pointnet.py:187-213, BatchNorm with batch statistics), differentiable in
    the parameters; the running statistics in `buffers` are updated in place when momentum >= 0."""

    @staticmethod
    def forward(ctx, cfg, x, attn, buffers, *params):
        num_anchors, zdim, reweight, eps, momentum, precision = cfg
        if len(params) != len(PNV2_PARAMS):
            raise ValueError(f"expected {len(PNV2_PARAMS)} parameter tensors")
        t = {n: _need(p.detach(), n) for n, p in zip(PNV2_PARAMS, params)}
        for n, b in zip(PNV2_BUFFERS, buffers):
            t[n] = _need(b.detach(), n)
        x = _need(x.detach().to(torch.float32).contiguous(), "x")
        attn = _need(attn.detach().to(device=x.device, dtype=torch.float32).contiguous(), "attn_weight")
        B, N, _ = x.shape
        prec = PRECISIONS[precision]
        lib = _ffi.lib()
        nbytes = lib.dfx_pointnet_v2_train_workspace_bytes(B, N, num_anchors, zdim)
        if nbytes == 0:
            raise ValueError(f"unsupported PointNetV2 training shape B={B} N={N} num_anchors={num_anchors}")
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        m = torch.empty(B, num_anchors, zdim, dtype=torch.float32, device=x.device)
        v = torch.empty_like(m)
        w = _pnv2_struct(t, num_anchors, zdim, reweight, eps)
        with torch.cuda.device(x.device):
            _ffi.check(lib.dfx_pointnet_v2_train_forward(ctypes.byref(w), ws_ptr, nbytes, x.data_ptr(), attn.data_ptr(), m.data_ptr(),
                                                         v.data_ptr(), float(momentum), B, N, prec, _ffi.current_stream()),
                       "dfx_pointnet_v2_train_forward")
        ctx.cfg, ctx.t, ctx.ws, ctx.ws_ptr, ctx.nbytes, ctx.attn, ctx.shape, ctx.prec = cfg, t, ws, ws_ptr, nbytes, attn, (B, N), prec
        ctx.leaves = [p if (p.is_leaf and p.requires_grad) else None for p in params]
        return m, v

    @staticmethod
    def backward(ctx, dm, dv):
        _check_first_backward(ctx, "PointNetV2TrainFn")
        num_anchors, zdim, reweight, eps, momentum, precision = ctx.cfg
        B, N = ctx.shape
        dm = _need(dm.contiguous(), "dm")
        dv = _need(dv.contiguous(), "dv")
        views = _flat_slices([ctx.t[n].shape for n in PNV2_PARAMS], dm.device)
        g = dict(zip(PNV2_PARAMS, views))
        w, gw = _pnv2_struct(ctx.t, num_anchors, zdim, reweight, eps), _pnv2_struct(g, num_anchors, zdim, reweight, eps)
        with torch.cuda.device(dm.device):
            _ffi.check(_ffi.lib().dfx_pointnet_v2_train_backward(ctypes.byref(w), ctx.ws_ptr, ctx.nbytes, ctx.attn.data_ptr(),
                                                                 dm.data_ptr(), dv.data_ptr(), ctypes.byref(gw), B, N, ctx.prec,
                                                                 _ffi.current_stream()), "dfx_pointnet_v2_train_backward")
        ctx.ws = None
        ctx.ws_ptr = None   # the workspace goes back to the allocator: a second backward must not reuse its address
        out = _assign_or_return(ctx.leaves, views)
        ctx.leaves = None
        return (None, None, None, None) + tuple(out)


def pointnet_v2_train_forward(params, buffers, x, attn, num_anchors=4, zdim=256, reweight_by_anchor=True, eps=1e-5, momentum=0.1,
                              precision="f32"):
    """`params` / `buffers`: dicts with the state_dict names of PointNetV2 (fp32 cuda tensors).  Returns (m, v)."""
    return PointNetV2TrainFn.apply((num_anchors, zdim, reweight_by_anchor, eps, momentum, precision), x, attn,
                                   [buffers[n] for n in PNV2_BUFFERS], *[params[n] for n in PNV2_PARAMS])


def flow_param_names(depth, n_class=4):
    """state_dict names (relative to the encoder) of the coupling flows, in libdfx's [part][layer][w0,b0,w1,b1,w2,b2] order."""
    return [f"flow.{i}.chain.{l}.net_s_t.{k}.{wb}" for i in range(n_class) for l in range(depth) for k in (0, 2, 4) for wb in ("weight", "bias")]


def _ptr_array(tensors):
    arr = (_ffi.c_fp * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = t.data_ptr()
    return arr


class PriorLossFn(torch.autograd.Function):
    """prior_loss = PartEncoder.get_prior_loss(part_code, ., logvar, valid)['prior_loss'] (part_encoders.py:1143-1182, use_flow),
    differentiable in part_code, logvar and the flow parameters."""

    @staticmethod
    def forward(ctx, cfg, part_code, logvar, valid, *params):
        depth, hidden, prior_var, kl_weight = cfg[:4]
        ctx.consumer = cfg[4] if len(cfg) > 4 else None     # see prior_loss(stream=...)
        if len(params) != 4 * depth * 6:
            raise ValueError(f"expected {4 * depth * 6} flow parameter tensors")
        ps = [_need(p.detach(), "flow parameter") for p in params]
        z = _need(part_code.detach().contiguous(), "part_code")
        lv = _need(logvar.detach().contiguous(), "logvar")
        B = z.shape[0]
        if z.shape != (B, 256, 4) or lv.shape != (B, 4, 256):
            raise ValueError("part_code (B,256,4) and logvar (B,4,256) expected")
        vd = valid.detach().to(device=z.device, dtype=torch.float32).contiguous()
        lib = _ffi.lib()
        nbytes = lib.dfx_prior_loss_workspace_bytes(B, depth, hidden)
        if nbytes == 0:
            raise ValueError(f"unsupported prior-loss configuration B={B} depth={depth} hidden={hidden}")
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=z.device)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        log_p = torch.empty(B, 4, dtype=torch.float32, device=z.device)
        ent = torch.empty(B, 4, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _ffi.check(lib.dfx_prior_loss_forward(_ptr_array(ps), depth, hidden, ws_ptr, nbytes, z.data_ptr(), lv.data_ptr(), vd.data_ptr(),
                                                  float(prior_var), float(kl_weight), loss.data_ptr(), log_p.data_ptr(), ent.data_ptr(), B,
                                                  _ffi.current_stream()), "dfx_prior_loss_forward")
        ctx.cfg, ctx.ps, ctx.ws, ctx.ws_ptr, ctx.nbytes, ctx.vd, ctx.B = cfg, ps, ws, ws_ptr, nbytes, vd, B
        ctx.leaves = [p if (p.is_leaf and p.requires_grad) else None for p in params]
        ctx.mark_non_differentiable(log_p, ent)
        return loss, log_p, ent

    @staticmethod
    def backward(ctx, g, _glp, _gent):
        _check_first_backward(ctx, "PriorLossFn")
        depth, hidden, prior_var, kl_weight = ctx.cfg[:4]
        B = ctx.B
        dev = ctx.vd.device
        views = _flat_slices([p.shape for p in ctx.ps], dev)
        dz = torch.empty(B, 256, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        dlv = torch.empty(B, 4, 256, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            _ffi.check(_ffi.lib().dfx_prior_loss_backward(_ptr_array(ctx.ps), depth, hidden, ctx.ws_ptr, ctx.nbytes, ctx.vd.data_ptr(),
                                                          float(prior_var), 1.0, _ptr_array(views), None if dz is None else dz.data_ptr(),
                                                          None if dlv is None else dlv.data_ptr(), B, _ffi.current_stream()),
                       "dfx_prior_loss_backward")
        # scale by the upstream scalar on the device (float(g) would drain the stream in the middle of the backward pass)
        views[0]._base.mul_(g)
        if dz is not None:
            dz.mul_(g)
        if dlv is not None:
            dlv.mul_(g)
        ctx.ws = None
        ctx.ws_ptr = None   # the workspace goes back to the allocator: a second backward must not reuse its address
        if ctx.consumer is not None:
            # autograd runs this node on the stream of its forward (the side stream) and orders dz / dlv for their consumers
            # itself; the parameter gradients are assigned by hand below, so the consuming stream is made to wait here
            side = torch.cuda.current_stream()
            if side != ctx.consumer:
                ctx.consumer.wait_stream(side)
                for t in (views[0]._base, dz, dlv):
                    if t is not None:
                        t.record_stream(ctx.consumer)
        out = _assign_or_return(ctx.leaves, views)
        ctx.leaves = None
        return (None, dz, dlv, None) + tuple(out)


_SIDE_STREAMS = {}


def side_stream(device):
    """One extra HIP stream per device for work that is independent of the main stream's (the prior-loss branch of the
    stage-1 step: ~170 few-row kernels that leave most of the chip idle, beside the denoiser's HBM-bound ones)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _SIDE_STREAMS[idx]


def prior_loss(flow_params, part_code, logvar, valid, depth=14, hidden=256, prior_var=1.0, kl_weight=5e-4, stream=None):
    """`flow_params`: dict with the encoder's state_dict names 'flow.{i}.chain.{l}.net_s_t.{0,2,4}.{weight,bias}'.
    Returns (prior_loss, log_p_part (B,4), entropy (B,4)).
    `stream`: run forward (and, through autograd, backward) on this stream instead of the current one.  The inputs are
    taken as of now (the stream waits for the current one); the CALLER makes the current stream wait for `stream` before
    it reads the outputs (`torch.cuda.current_stream().wait_stream(stream)`); the backward pass needs nothing from the
    caller: gradients are handed back in stream order."""
    names = flow_param_names(depth)
    if stream is None or stream == torch.cuda.current_stream(part_code.device):
        return PriorLossFn.apply((depth, hidden, prior_var, kl_weight), part_code, logvar, valid, *[flow_params[n] for n in names])
    main = torch.cuda.current_stream(part_code.device)
    stream.wait_stream(main)
    with torch.cuda.stream(stream):
        out = PriorLossFn.apply((depth, hidden, prior_var, kl_weight, main), part_code, logvar, valid, *[flow_params[n] for n in names])
    for t in (part_code, logvar, valid):
        t.record_stream(stream)
    for t in out:
        t.record_stream(main)
    return out


ALIGNER_TOP = {"proj_in_w": "proj_in.weight", "proj_in_b": "proj_in.bias", "class_emb": "class_emb.weight", "post_norm_w": "post_norm.weight",
               "post_norm_b": "post_norm.bias", "proj_out_w": "proj_out.weight", "proj_out_b": "proj_out.bias"}
ALIGNER_BLOCK = {"norm2_w": "norm2.weight", "norm2_b": "norm2.bias", "to_q": "attn2.to_q.weight", "to_k": "attn2.to_k.weight", "to_v": "attn2.to_v.weight",
                 "to_out_w": "attn2.to_out.0.weight", "to_out_b": "attn2.to_out.0.bias", "norm3_w": "norm3.weight", "norm3_b": "norm3.bias",
                 "ff_proj_w": "ff.net.0.proj.weight", "ff_proj_b": "ff.net.0.proj.bias", "ff_out_w": "ff.net.2.weight", "ff_out_b": "ff.net.2.bias"}


def aligner_param_names(depth):
    """state_dict names (relative to the part aligner) of the parameters the training kernels differentiate, in a fixed order.  ``pre_norm.*``
    exists in the state_dict but is unused with cimle / cond_noise_type 0 (part_encoders.py:119-131): no gradient, like under torch autograd."""
    return list(ALIGNER_TOP.values()) + [f"transformer_blocks.{i}.{k}" for i in range(depth) for k in ALIGNER_BLOCK.values()]


def _aligner_struct(tensors, cfg):
    n_class, zdim, n_heads, d_head, noise_dim, noise_scale, depth = cfg
    w = _ffi.LatentWeights()
    w.n_class, w.zdim, w.flow_depth, w.flow_hidden, w.depth, w.n_heads, w.d_head = n_class, zdim, 0, 0, depth, n_heads, d_head
    w.cimle, w.noise_dim, w.noise_scale, w.prior_var, w.log_scale_var = 1, noise_dim, float(noise_scale), 1.0, 0.0
    it = iter(tensors)
    for field in ALIGNER_TOP:
        setattr(w, field, next(it).data_ptr())
    for i in range(depth):
        for field in ALIGNER_BLOCK:
            setattr(w.blocks[i], field, next(it).data_ptr())
    return w


class AlignerTrainFn(torch.autograd.Function):
    """PartAlignerTransformer.forward (part_encoders.py:88-143; the shipped cIMLE configuration) with its backward on libdfx's exact-fp32 training
    kernels (aligner_train.hip): differentiable in the aligner's parameters and in part_code."""

    @staticmethod
    def forward(ctx, cfg, part_code, valid, noise, *params):
        n_class, zdim, n_heads, d_head, noise_dim, noise_scale, depth = cfg
        ps = [_need(p.detach(), "aligner parameter") for p in params]
        z = _need(part_code.detach().to(torch.float32).contiguous(), "part_code")
        B = z.shape[0]
        if tuple(z.shape) != (B, zdim, n_class):
            raise ValueError(f"part_code: expected (B, {zdim}, {n_class}), got {tuple(z.shape)}")
        vd = valid.detach().to(device=z.device, dtype=torch.float32).contiguous()
        nz = noise.detach().to(device=z.device, dtype=torch.float32).contiguous()
        if tuple(vd.shape) != (B, n_class) or tuple(nz.shape) != (B, noise_dim):
            raise ValueError("valid (B, n_class) and noise (B, noise_dim) expected")
        lib = _ffi.lib()
        nbytes = lib.dfx_aligner_train_workspace_bytes(B, n_class, zdim, noise_dim, n_heads, d_head, depth)
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=z.device)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        mean = torch.empty(B, 3, n_class, dtype=torch.float32, device=z.device)
        logvar = torch.empty_like(mean)
        w = _aligner_struct(ps, cfg)
        with torch.cuda.device(z.device):
            _ffi.check(lib.dfx_aligner_train_forward(w, ws_ptr, nbytes, z.data_ptr(), vd.data_ptr(), nz.data_ptr(), mean.data_ptr(), logvar.data_ptr(), B,
                                                     _ffi.current_stream()), "dfx_aligner_train_forward")
        ctx.cfg, ctx.ps, ctx.ws, ctx.ws_ptr, ctx.nbytes, ctx.vd, ctx.B = cfg, ps, ws, ws_ptr, nbytes, vd, B
        ctx.leaves = [p if (p.is_leaf and p.requires_grad) else None for p in params]
        return mean, logvar

    @staticmethod
    def backward(ctx, d_mean, d_logvar):
        _check_first_backward(ctx, "AlignerTrainFn")
        n_class, zdim = ctx.cfg[0], ctx.cfg[1]
        dev = ctx.vd.device
        views = _flat_slices([p.shape for p in ctx.ps], dev)
        dz = torch.empty(ctx.B, zdim, n_class, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        dm = None if d_mean is None else _need(d_mean.contiguous(), "d_mean")
        dl = None if d_logvar is None else _need(d_logvar.contiguous(), "d_logvar")
        w, g = _aligner_struct(ctx.ps, ctx.cfg), _aligner_struct(views, ctx.cfg)
        with torch.cuda.device(dev):
            _ffi.check(_ffi.lib().dfx_aligner_train_backward(w, ctx.ws_ptr, ctx.nbytes, ctx.vd.data_ptr(), _ffi.ptr(dm), _ffi.ptr(dl), g, _ffi.ptr(dz), ctx.B,
                                                             _ffi.current_stream()), "dfx_aligner_train_backward")
        ctx.ws = None
        ctx.ws_ptr = None
        out = _assign_or_return(ctx.leaves, views)
        ctx.leaves = None
        return (None, dz, None, None) + tuple(out)


def aligner_train_forward(params, part_code, valid, noise, n_class=4, zdim=256, n_heads=8, d_head=32, noise_dim=32, noise_scale=100.0):
    """`params`: dict state_dict-key (relative to the part aligner) -> fp32 cuda tensor.  Returns (mean, logvar), each (B, 3, n_class), differentiable in
    the parameters and in part_code (valid / noise are data)."""
    depth = 0
    while f"transformer_blocks.{depth}.norm2.weight" in params:
        depth += 1
    cfg = (n_class, zdim, n_heads, d_head, noise_dim, float(noise_scale), depth)
    return AlignerTrainFn.apply(cfg, part_code, valid, noise, *[params[n] for n in aligner_param_names(depth)])


def dropout_factors(seed, site, p, n, device="cuda"):
    """The factors (0 or 1/(1-p)) the training kernels apply to `n` consecutive elements of a dropout site (tests / debugging):
    site 2 i = behind to_out of block i over (B N, 128); 2 i + 1 = behind the GEGLU of block i over (B N, 512); 1000 =
    time_embed over (B, 1024)."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _ffi.check(_ffi.lib().dfx_debug_dropout_factors(int(seed) & (2 ** 64 - 1), int(site), float(p), out.data_ptr(), n,
                                                        _ffi.current_stream()), "dfx_debug_dropout_factors")
    return out


class MaskedMSEFn(torch.autograd.Function):
    """((target - pred)^2 * flags).mean(1).sum() / flags.sum()  (anchored_diffusion.py:840-847); differentiable in pred."""

    @staticmethod
    def forward(ctx, target, pred, flags):
        target = _need(target.detach().contiguous(), "target")
        predc = _need(pred.detach().contiguous(), "pred")
        B, _, N = predc.shape
        fl = None if flags is None else _need(flags.detach().reshape(B, N).to(torch.float32).contiguous(), "flags")
        ws2 = torch.zeros(2, dtype=torch.float64, device=predc.device)
        loss = torch.empty((), dtype=torch.float32, device=predc.device)
        with torch.cuda.device(predc.device):
            _ffi.check(_ffi.lib().dfx_masked_mse_f32(target.data_ptr(), predc.data_ptr(), None if fl is None else fl.data_ptr(),
                                                     ws2.data_ptr(), loss.data_ptr(), B, N, _ffi.current_stream()),
                       "dfx_masked_mse_f32")
        ctx.saved = (target, predc, fl, ws2)
        return loss

    @staticmethod
    def backward(ctx, g):
        target, pred, fl, ws2 = ctx.saved
        B, _, N = pred.shape
        d = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            _ffi.check(_ffi.lib().dfx_masked_mse_backward_f32(target.data_ptr(), pred.data_ptr(),
                                                              None if fl is None else fl.data_ptr(), ws2.data_ptr(), 1.0,
                                                              d.data_ptr(), B, N, _ffi.current_stream()),
                       "dfx_masked_mse_backward_f32")
        # the upstream scalar stays on the device: reading it (float(g)) would stall the host until the whole forward has run
        return None, d.mul_(g), None


def masked_mse(target, pred, flags=None):
    return MaskedMSEFn.apply(target, pred, flags)


def generation_of(tensors):
    """Generation of a parameter SET: `Adam.step` updates parameters through raw pointers, which torch's in-place version counters do
    not see, so it bumps a counter on every tensor it steps (`_dfx_gen`); caches of packed weights (modules.TransformerNet.engine,
    encoders.PointNetV2 / the latent samplers) key on the sum over THEIR OWN parameters — stepping one parameter set (say the stage-2
    modules) leaves the engines and handles of every other set (a frozen stage-1 denoiser, an evaluation copy) alone.
    The counter also lives per STORAGE (`_STORAGE_GEN`, keyed by the storage's base address): a parameter updated through an alias
    — a second Parameter or a view on the same memory, tensors re-wrapped after the flat re-pointing — still invalidates the caches of
    every tensor object that shares that memory."""
    return sum(getattr(t, "_dfx_gen", 0) + _STORAGE_GEN.get(_storage_key(t), 0) for t in tensors)


_STORAGE_GEN = {}


def _storage_key(t):
    try:
        return (t.device.type, t.device.index, t.untyped_storage().data_ptr())
    except Exception:   # noqa: BLE001 - tensors without storage (meta, fake): object identity only
        return None


def _bump_generation(tensors):
    seen = set()
    for t in tensors:
        t._dfx_gen = getattr(t, "_dfx_gen", 0) + 1
        k = _storage_key(t)
        if k is not None and k not in seen:
            seen.add(k)
            _STORAGE_GEN[k] = _STORAGE_GEN.get(k, 0) + 1


class Adam:
    """torch.optim.Adam (amsgrad off) with Runner.train's clip_grad_norm_(max_norm) in front of it (runner.py:312-316).
    `params`: iterable of fp32 cuda tensors with .grad set by backward().  When the gradients sit back to back in one
    buffer — as DenoiserTrainFn.backward leaves them — the parameters are re-pointed (once, at the first step; values
    unchanged, p.data becomes a view) into ONE flat buffer in the same order, next to flat moment buffers, and the
    gradient norm and the update are one kernel launch each instead of three per tensor (flatten=False: never)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=10.0, flatten=True):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.step_count = 0
        self.flatten = flatten
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self._flat = None            # (order, offsets, flat_p, flat_m, flat_v) once the layout of the gradients is known
        self.last_step_was_flat = False
        dev = self.params[0].device
        self._ws = torch.zeros(1024, dtype=torch.float64, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _grad_groups(self):
        """Partition the parameters by the storage their gradient lives in.  A group whose gradients are disjoint fp32
        slices of ONE buffer (gaps allowed: they hold zeros, as the libdfx backward functions leave them) is returned as
        (order, offsets, total); parameters whose gradient stands alone come back in `loose`.  None if a gradient is missing."""
        if any(p.grad is None for p in self.params):
            return None
        by_store = {}
        for i, p in enumerate(self.params):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda:
                by_store.setdefault(("loose", i), []).append(i)
            else:
                by_store.setdefault(g.untyped_storage().data_ptr(), []).append(i)
        groups, loose = [], []
        for key, idxs in by_store.items():
            if isinstance(key, tuple) or len(idxs) == 1:
                loose += idxs
                continue
            order = sorted(idxs, key=lambda i: self.params[i].grad.data_ptr())
            g0 = self.params[order[0]].grad
            base, store = g0.data_ptr(), g0.untyped_storage()
            offsets, end, ok = [], 0, base == store.data_ptr()
            for i in order:
                g = self.params[i].grad
                off = (g.data_ptr() - base) // 4
                if (g.data_ptr() - base) % 4 or off < end:
                    ok = False
                    break
                offsets.append(off)
                end = off + g.numel()
            if ok and end * 4 <= store.nbytes():
                groups.append((order, offsets, end))
            else:
                loose += idxs
        return groups, loose

    def _flat_work(self):
        """[(flat_param, flat_grad, flat_m, flat_v)] for the grouped parameters + [(p, g, m, v)] for the loose ones, or None."""
        if not self.flatten:
            return None
        lay = self._grad_groups()
        if lay is None:
            return None
        groups, loose = lay
        if self._flat is None:
            self._flat = {}
        work = []
        dev = self.params[0].device
        for order, offsets, total in groups:
            key = (tuple(order), tuple(offsets), total)
            if key not in self._flat:
                fp, fm, fv = (torch.zeros(total, dtype=torch.float32, device=dev) for _ in range(3))
                with torch.no_grad():
                    for i, o in zip(order, offsets):
                        p, k = self.params[i], self.params[i].numel()
                        fp[o:o + k].copy_(p.detach().reshape(-1))
                        fm[o:o + k].copy_(self.m[i].reshape(-1))
                        fv[o:o + k].copy_(self.v[i].reshape(-1))
                        p.data = fp[o:o + k].view_as(p)
                        self.m[i] = fm[o:o + k].view_as(p)
                        self.v[i] = fv[o:o + k].view_as(p)
                self._flat[key] = (fp, fm, fv)
            fp, fm, fv = self._flat[key]
            g0 = self.params[order[0]].grad
            work.append((fp, torch.as_strided(g0, (total,), (1,)), fm, fv))   # same storage, layout checked above
        self.last_step_was_flat = len(groups) > 0 and not loose
        for i in loose:
            work.append((self.params[i], _need(self.params[i].grad.contiguous(), "grad"), self.m[i], self.v[i]))
        return work

    def grad_norm(self):
        """Global L2 norm of the gradients (what clip_grad_norm_ returns), as a 0-dim float64 cuda tensor."""
        lib = _ffi.lib()
        self._sumsq.zero_()
        work = self._flat_work()
        todo = [w[1] for w in work] if work is not None else [p.grad for p in self.params if p.grad is not None]
        for g in todo:
            g = _need(g.contiguous(), "grad")
            with torch.cuda.device(g.device):
                _ffi.check(lib.dfx_grad_sumsq_accumulate(g.data_ptr(), g.numel(), self._ws.data_ptr(), self._sumsq.data_ptr(),
                                                         _ffi.current_stream()), "dfx_grad_sumsq_accumulate")
        return self._sumsq.sqrt()[0]

    @torch.no_grad()
    def step(self):
        lib = _ffi.lib()
        norm = self.grad_norm() if self.max_norm and self.max_norm > 0 else None
        self.step_count += 1
        _bump_generation(self.params)
        work = self._flat_work()
        if work is None:
            self.last_step_was_flat = False
            work = [(p, _need(p.grad.contiguous(), "grad"), m, v) for p, m, v in zip(self.params, self.m, self.v) if p.grad is not None]
        for p, g, m, v in work:
            with torch.cuda.device(p.device):
                _ffi.check(lib.dfx_adam_step_f32(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                                 self._sumsq.data_ptr() if norm is not None else None,
                                                 float(self.max_norm or 0.0), float(self.lr), float(self.betas[0]),
                                                 float(self.betas[1]), float(self.eps), float(self.weight_decay),
                                                 self.step_count, _ffi.current_stream()), "dfx_adam_step_f32")
        return norm


def linear_lr(epoch, start_epoch, end_epoch, start_lr, end_lr):
    """The reference's `LinearLR` schedule (optimizers/schedulers.py:8-19, a LambdaLR factor on the optimiser's base lr =
    start_lr): constant up to start_epoch, linear to end_lr at end_epoch, constant after.  Returns the learning rate."""
    if epoch <= start_epoch:
        f = 1.0
    elif epoch <= end_epoch:
        frac = (epoch - start_epoch) / (end_epoch - start_epoch)
        f = (1 - frac) * 1.0 + frac * (end_lr / start_lr)
    else:
        f = end_lr / start_lr
    return start_lr * f


class _PinnedRing:
    """Small host arrays go to the device through a ring of pinned buffers with a non-blocking copy: `tensor.to(device)` from pageable
    memory makes the host wait until everything queued on the stream before the copy has run — in the middle of a training step that
    is a full drain of the encoder's forward, and the device then idles while the host catches up with the denoiser's launches."""

    def __init__(self, slots=4):
        self.slots, self.buf, self.ev, self.i = slots, {}, {}, 0

    def to_device(self, array, device):
        src = torch.from_numpy(array)
        key = (tuple(src.shape), src.dtype)
        if key not in self.buf:
            self.buf[key] = [torch.empty(src.shape, dtype=src.dtype).pin_memory() for _ in range(self.slots)]
            self.ev[key] = [None] * self.slots
        k = self.i % self.slots
        self.i += 1
        if self.ev[key][k] is not None:
            self.ev[key][k].synchronize()      # the copy that last used this slot (several steps ago) has long run
        self.buf[key][k].copy_(src)
        out = self.buf[key][k].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(out.device))
        self.ev[key][k] = ev
        return out


_pinned = _PinnedRing()


def stage1_losses(encoder, diffusion, pcds, device="cuda", epoch=0, t=None, diffusion_loss_weight=1.0, noise=None, overlap_prior=True, detach_anchor=True):
    """The training forward of the reference's agent for stage 1 (AnchorDiffAE.forward, anchor_gen.py:970-1020): the encoder's
    training forward (part codes, prior loss, ground-truth anchors per point, ctx), one timestep per shape, and the denoiser's
    masked MSE.  `encoder` / `diffusion`: difffacto_amd.encoders.PartEncoderForTransformerDecoder / modules.AnchoredDiffusion in
    train() mode; `t`: (B,) timesteps (the reference draws them with its Uniform sampler, samplers/sampler.py:25-40); `noise`: the
    diffusion noise (B,3,N) or None.  Returns the loss dict; sum the entries whose key contains 'loss' and call backward().
    overlap_prior: the prior loss (flows) runs on a second stream beside the denoiser, forward and backward; results are the same.
    The same function serves stage 2 (an encoder with a part aligner: fit_loss in the dict, gradients to the aligner through ctx[1] and through
    the per-point variance); the name is historical."""
    import numpy as np
    ref = pcds["ref"].to(device)
    seg = pcds["ref_seg_mask"].to(device).to(torch.int32)
    B = ref.shape[0]
    side = side_stream(ref.device) if overlap_prior else None
    encoder.prior_loss_stream = side      # the prior-loss branch does not feed the denoiser: it runs beside it
    try:
        ctx, mean_pp, logvar_pp, _flag_pp, losses, _ = encoder(pcds, device, epoch=epoch)
    finally:
        encoder.prior_loss_stream = None
    variance_pp = torch.exp(logvar_pp)     # BEFORE the detaches, like anchor_gen.py:1002: with a part aligner (stage 2) it carries the gradient to logvar
    if not detach_anchor:
        raise NotImplementedError("detach_anchor=False (a gradient through the per-point anchors) is not implemented natively")
    mean_pp = mean_pp.detach()                                                                                 # anchor_gen.py:1011-1012
    if t is None:
        t = _pinned.to_device(np.random.choice(diffusion.num_timesteps, size=(B,)), device)   # (host-drawn like the reference's sampler)
    dp = pcds.get("dp_present", None)
    flags = None
    if dp is not None:
        dp = dp.to(device).to(torch.float32)
        flags = torch.gather(dp[:, None, :], 2, seg.long()[:, None, :])
    d = diffusion.training_losses(ref.transpose(1, 2).contiguous(), t, anchors=mean_pp, variance=variance_pp, ctx=ctx,
                                  anchor_assignment=seg, valid_id=dp, flags=flags, noise=noise)
    if side is not None:
        torch.cuda.current_stream(ref.device).wait_stream(side)      # from here on the encoder's loss entries may be read
    losses = dict(losses)
    losses["mse_loss"] = diffusion_loss_weight * d["mse_loss"]
    return losses

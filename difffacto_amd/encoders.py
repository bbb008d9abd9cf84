"""Host-side mirrors of the reference's gen-path encoder API, backed by libdfx.

* ``CouplingLayer`` / ``SequentialFlow`` / ``build_latent_flow``  parameter containers with the reference's
  attribute names (python/difffacto/models/encoders/flow.py:7-79)
* ``PartAlignerTransformer``   constructor arguments, parameter names and ``forward(x, mask, noise)`` of
  ``ENCODERS['PartAlignerTransformer']`` (python/difffacto/models/encoders/part_encoders.py:19-143)
* ``PartEncoderForTransformerDecoder``  the generation entry point ``sample_latents`` with the reference's
  signature and 6-tuple result (part_encoders.py:1052-1110, prepare_ctx :1317-1326)
* ``generate``   the ``not self.training and self.gen`` branch of ``AnchorDiffAE.forward``
  (python/difffacto/models/networks/anchor_gen.py:1034-1084): sample_latents -> decode

All math runs in libdfx (``dfx_sample_latents`` / ``dfx_part_aligner`` / ``dfx_flow_reverse``); the modules only hold
parameters under the reference's ``state_dict`` keys and draw the random inputs with ``torch.randn`` exactly where
the reference does.  Option combinations outside the shipped ``configs/gen_*.py`` raise ``NotImplementedError``.
``PointNetV2`` (the encode-side part encoder) is native in inference as well; ``PointNet2SSG`` / ``PointNet2MSG``
(pointnet2.py, round 6) compose the mirrored set-abstraction layers.
"""
import ctypes
import math
import weakref

import torch
import torch.nn as nn

from . import _ffi
from .latents import LatentSampler
from .modules import _BlockParams, decode


def _unsupported(what):
    raise NotImplementedError(f"libdfx implements the shipped gen_* encoder configuration only: {what}")



def _training_generation(tensors):
    """training.Adam updates parameters through raw pointers (no torch version bump): packed-weight caches key on the generation of
    their own parameter set (training.generation_of)."""
    from . import training
    return training.generation_of(tensors)


class PointNetV2(nn.Module):
    """``ENCODERS['PointNetV2']`` (python/difffacto/models/encoders/pointnet.py:124-213): per-point MLP, attention-weighted
    max-pool per part, per-part heads -> (m, v) of shape (B, num_anchors, zdim).  Same constructor arguments and parameter
    names.  In inference (eval + no_grad) the whole forward is ``dfx_pointnet_v2_forward_f32``; in train() mode forward and
    backward are ``dfx_pointnet_v2_train_forward / _backward`` (batch-statistics BatchNorm); only eval-mode statistics WITH
    autograd fall back to the module's own torch layers."""

    def __init__(self, point_dim=3, zdim=1024, num_anchors=4, reweight_by_anchor=True, use_ln=False, per_part_mlp=False):
        super().__init__()
        if use_ln or not per_part_mlp or point_dim != 3:
            _unsupported("PointNetV2 needs per_part_mlp=True, use_ln=False, point_dim=3 (configs/gen_*.py)")
        self.reweight_by_anchor, self.per_part_mlp, self.zdim, self.num_anchors, self.use_ln = \
            reweight_by_anchor, per_part_mlp, zdim, num_anchors, use_ln
        widths = (point_dim, 128, 128, 256, 512)
        for i in range(4):
            setattr(self, f"conv{i + 1}", nn.Conv1d(widths[i], widths[i + 1], 1))
        for i in range(4):
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(widths[i + 1]))
        A = num_anchors

        def head():
            return nn.Sequential(nn.Conv1d(512 * A, 256 * A, 1, groups=A), nn.BatchNorm1d(256 * A), nn.ReLU(),
                                 nn.Conv1d(256 * A, 128 * A, 1, groups=A), nn.BatchNorm1d(128 * A), nn.ReLU(),
                                 nn.Conv1d(128 * A, zdim * A, 1, groups=A))
        self.mlp_m, self.mlp_v = head(), head()

    # ---- libdfx handle, rebuilt when a parameter / buffer changes ----
    def _handle(self):
        ts = [t for t in list(self.parameters()) + list(self.buffers()) if t.dtype == torch.float32]
        ver = (_training_generation(ts),) + tuple((t._version, t.data_ptr()) for t in ts)
        if self.__dict__.get("_h") is None or self.__dict__.get("_ver") != ver:
            self._close()
            keep = []

            def dp(t):
                t = t.detach().to(torch.float32).contiguous()
                keep.append(t)
                return t.data_ptr()

            w = _ffi.PointNetV2Weights()
            w.num_anchors, w.zdim, w.reweight_by_anchor, w.bn_eps = self.num_anchors, self.zdim, int(self.reweight_by_anchor), self.bn1.eps
            for i in range(4):
                conv, bn = getattr(self, f"conv{i + 1}"), getattr(self, f"bn{i + 1}")
                w.conv_w[i], w.conv_b[i] = dp(conv.weight), dp(conv.bias)
                w.bn_w[i], w.bn_b[i], w.bn_mean[i], w.bn_var[i] = dp(bn.weight), dp(bn.bias), dp(bn.running_mean), dp(bn.running_var)
            for k, head in enumerate((self.mlp_m, self.mlp_v)):
                for l, ci in enumerate((0, 3, 6)):
                    w.head_w[k][l], w.head_b[k][l] = dp(head[ci].weight), dp(head[ci].bias)
                for l, bi in enumerate((1, 4)):
                    bn = head[bi]
                    w.head_bn_w[k][l], w.head_bn_b[k][l] = dp(bn.weight), dp(bn.bias)
                    w.head_bn_mean[k][l], w.head_bn_var[k][l] = dp(bn.running_mean), dp(bn.running_var)
            h = ctypes.c_void_p()
            with torch.cuda.device(self.conv1.weight.device):
                rc = _ffi.lib().dfx_pointnet_v2_create(ctypes.byref(h), ctypes.byref(w), _ffi.current_stream())
            _ffi.check(rc, "dfx_pointnet_v2_create")
            self.__dict__["_h"], self.__dict__["_ver"] = h, ver
        return self.__dict__["_h"]

    def _close(self):
        if self.__dict__.get("_h") is not None:
            _ffi.lib().dfx_pointnet_v2_destroy(self.__dict__["_h"])
            self.__dict__["_h"] = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def forward(self, x, attn_weight):
        """x (B,N,3), attn_weight (B,N,num_anchors) -> (m, v), each (B, num_anchors, zdim)."""
        B, N, _ = x.shape
        A = self.num_anchors
        # eval(): the native forward runs whenever no gradient can be asked for through it — under no_grad, and also with autograd
        # enabled when neither an input nor a parameter requires grad (a frozen encoder called outside no_grad, like the
        # reference's nn.Module would simply run)
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or attn_weight.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        if not self.training and not wants_grad:
            if not x.is_cuda:
                raise RuntimeError("PointNetV2: CPU not supported")
            x = x.detach().to(torch.float32).contiguous()
            attn_weight = attn_weight.detach().to(device=x.device, dtype=torch.float32).contiguous()
            m = torch.empty(B, A, self.zdim, dtype=torch.float32, device=x.device)
            v = torch.empty_like(m)
            with torch.cuda.device(x.device):
                rc = _ffi.lib().dfx_pointnet_v2_forward_f32(self._handle(), _ffi.ptr(x), _ffi.ptr(attn_weight), _ffi.ptr(m), _ffi.ptr(v),
                                                            B, N, _ffi.current_stream())
            _ffi.check(rc, "dfx_pointnet_v2_forward_f32")
            return m, v
        if self.training and x.is_cuda and B >= 2 and A == 4 and self.zdim % 4 == 0:
            # train() mode: batch-statistics BatchNorm forward (running statistics updated in place like nn.BatchNorm1d) and
            # the backward on libdfx's training kernels (difffacto_amd/training.py).  Exact fp32 unless the caller sets
            # `module.train_precision = "bf16"` (bf16 operands for the trunk's products: the max-pool's arg-max may flip under
            # that noise, so results then differ from fp32 discontinuously)
            from . import training as _training
            sd_p, sd_b = dict(self.named_parameters()), dict(self.named_buffers())
            momentum = self.bn1.momentum if self.bn1.momentum is not None else 0.1
            m, v = _training.pointnet_v2_train_forward(sd_p, sd_b, x, attn_weight, num_anchors=A, zdim=self.zdim,
                                                       reweight_by_anchor=self.reweight_by_anchor, eps=self.bn1.eps, momentum=momentum,
                                                       precision=getattr(self, "train_precision", "f32"))
            with torch.no_grad():   # (one launch for the eight counters instead of eight)
                nbt = [mod.num_batches_tracked for mod in self.modules() if isinstance(mod, nn.BatchNorm1d) and mod.num_batches_tracked is not None]
                if nbt:
                    torch._foreach_add_(nbt, 1)
            self.__dict__["_ver"] = None      # the running statistics changed under the eval handle's feet
            return m, v
        # No second implementation behind the native one: what libdfx does not run raises, like every other
        # unsupported option of this module.
        if not x.is_cuda:
            raise RuntimeError("PointNetV2: CPU not supported")
        if not self.training:
            _unsupported("PointNetV2 in eval() mode with a gradient required through it (an input or a parameter has requires_grad; "
                         "running-statistics BatchNorm has no native backward): freeze the module (requires_grad_(False)), wrap the "
                         "call in torch.no_grad(), or call .train() for the stage-1 training step")
        _unsupported(f"PointNetV2.train() needs batch >= 2 (BatchNorm batch statistics), num_anchors == 4 and zdim % 4 == 0; "
                     f"got B={B}, num_anchors={A}, zdim={self.zdim}")


class PointNet2SSG(nn.Module):
    """``ENCODERS['PointNet2SSG']`` (python/difffacto/models/encoders/pointnet2.py:7-79): three set-abstraction layers (2048 -> 512 centres, r = 0.2, 64
    neighbours; -> 128, r = 0.4; -> global) and a Linear / BatchNorm1d / ReLU / Dropout head -> (B, num_anchors, zdim).  Same constructor (the reference's
    spelling ``additioinal_dim`` included), same ``state_dict`` keys (``SA_modules.{i}.mlps.{k}.{j}.*``, ``fc_layer.{0,1,3,4,7}.*``).  Registered by the
    reference, selected by none of the shipped configs.  Everything runs on libdfx in every mode: FPS / ball query / grouping and the shared MLPs through the
    mirrored ``PointnetSAModule`` (eval: the fused gather + MLP + max kernels; train() / autograd: ``dfx_shared_mlp_train_*``), the head through the same
    training kernels (rows = the batch: BatchNorm1d over B); ``nn.Dropout(0.5)`` is torch's elementwise op on torch's generator, as in the reference."""

    def __init__(self, additioinal_dim=4, zdim=256, num_anchors=4):
        super().__init__()
        self.additioinal_dim, self.zdim, self.num_anchors = additioinal_dim, zdim, num_anchors
        self._build_model()

    def _build_model(self):
        from .pointnet2_ops.pointnet2_modules import PointnetSAModule
        self.SA_modules = nn.ModuleList([
            PointnetSAModule(npoint=512, radius=0.2, nsample=64, mlp=[self.additioinal_dim, 64, 64, 128], use_xyz=True),
            PointnetSAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256], use_xyz=True),
            PointnetSAModule(mlp=[256, 256, 512, 1024], use_xyz=True)])
        self.fc_layer = nn.Sequential(nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.ReLU(True), nn.Linear(512, 256, bias=False),
                                      nn.BatchNorm1d(256), nn.ReLU(True), nn.Dropout(0.5), nn.Linear(256, self.zdim * self.num_anchors))

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def _head(self, feat):
        """fc_layer on libdfx: (B, 1024) -> (B, zdim * num_anchors).  The rows of the training kernels are the B shapes (x as (1, C, B, 1))."""
        from .pointnet2_ops.pointnet2_modules import mlp_train
        fc = self.fc_layer
        if fc[1].training != fc[4].training:
            _unsupported("PointNet2 head with its two BatchNorm1d layers in different modes")
        x = feat.t().contiguous()[None, :, :, None]
        h = mlp_train([(fc[0], fc[1]), (fc[3], fc[4])], x, pool=False)                   # Linear + BatchNorm1d + ReLU, twice
        h = fc[6](h)                                                                       # nn.Dropout(0.5): identity in eval()
        return mlp_train([(fc[7], None)], h, pool=False, relu_mask=0)[0, :, :, 0].t()      # Linear with bias

    def forward(self, pointcloud):
        if not pointcloud.is_cuda:
            raise RuntimeError("PointNet2 encoder: CPU not supported")
        B = pointcloud.shape[0]
        xyz, features = self._break_up_pc(pointcloud)
        for module in self.SA_modules:
            xyz, features = module(xyz, features)
        return self._head(features.squeeze(-1)).reshape(B, self.num_anchors, self.zdim)


class PointNet2MSG(PointNet2SSG):
    """``ENCODERS['PointNet2MSG']`` (pointnet2.py:82-115): the multi-scale-grouping variant (three radii per layer, channel concat)."""

    def _build_model(self):
        super()._build_model()
        from .pointnet2_ops.pointnet2_modules import PointnetSAModule, PointnetSAModuleMSG
        a = self.additioinal_dim
        c1 = 64 + 128 + 128
        self.SA_modules = nn.ModuleList([
            PointnetSAModuleMSG(npoint=512, radii=[0.1, 0.2, 0.4], nsamples=[16, 32, 128], mlps=[[a, 32, 32, 64], [a, 64, 64, 128], [a, 64, 96, 128]], use_xyz=True),
            PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4, 0.8], nsamples=[32, 64, 128],
                                mlps=[[c1, 64, 64, 128], [c1, 128, 128, 256], [c1, 128, 128, 256]], use_xyz=True),
            PointnetSAModule(mlp=[128 + 256 + 256, 256, 512, 1024], use_xyz=True)])


class CouplingLayer(nn.Module):
    def __init__(self, d, intermediate_dim, swap=False):
        super().__init__()
        if d % 2:
            _unsupported("odd latent_dim in CouplingLayer")
        self.d = d - (d // 2)
        self.swap = swap
        self.net_s_t = nn.Sequential(nn.Linear(self.d, intermediate_dim), nn.ReLU(inplace=True),
                                     nn.Linear(intermediate_dim, intermediate_dim), nn.ReLU(inplace=True),
                                     nn.Linear(intermediate_dim, (d - self.d) * 2))      # flow.py:13-19


class SequentialFlow(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.chain = nn.ModuleList(layers)                                               # flow.py:54-56


def build_latent_flow(latent_flow_depth, latent_flow_hidden_dim, latent_dim):
    return SequentialFlow([CouplingLayer(latent_dim, latent_flow_hidden_dim, swap=(i % 2 == 0))
                           for i in range(latent_flow_depth)])                           # flow.py:75-79


class PartAlignerTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, out_channels, depth=1, dropout=0., use_linear=False, n_class=4,
                 use_checkpoint=False, single_attn=False, class_cond=True, mask_out_unreferenced_code=True, cimle=False,
                 noise_dim=32, noise_scale=10, cimle_start_epoch=0, add_class_cond=False, cond_noise_type=0,
                 cond_noise_as_token=False):
        super().__init__()
        if not (use_linear and single_attn and class_cond and add_class_cond and mask_out_unreferenced_code):
            _unsupported("PartAlignerTransformer needs use_linear, single_attn, class_cond, add_class_cond, "
                         "mask_out_unreferenced_code")
        if cond_noise_as_token or (cimle and cond_noise_type != 0):
            _unsupported(f"cond_noise_type={cond_noise_type}")
        if out_channels != 6:
            _unsupported("out_channels != 6")
        self.n_class, self.cimle, self.noise_scale, self.noise_dim = n_class, cimle, noise_scale, noise_dim
        self.cimle_start_epoch, self.cond_noise_type = cimle_start_epoch, cond_noise_type
        self.zdim = in_channels
        self.n_heads, self.d_head = n_heads, d_head
        self.in_channels = in_channels + int(cimle) * noise_dim                          # part_encoders.py:46
        inner = n_heads * d_head
        self.inner_dim = inner
        self.class_emb = nn.Embedding(n_class, inner)
        self.pre_norm = nn.LayerNorm(inner)
        self.post_norm = nn.LayerNorm(inner)
        self.proj_in = nn.Linear(self.in_channels, inner)
        self.transformer_blocks = nn.ModuleList([_BlockParams(inner, n_heads, d_head, inner, dropout) for _ in range(depth)])
        self.proj_out = nn.Linear(inner, out_channels)

    def forward(self, x, mask=None, noise=None):
        """(B,zdim,n_class), (B,n_class), (B,noise_dim) -> mean (B,3,n_class), logvar (B,3,n_class).  With a gradient required through it (stage 2:
        train_*_stage2.py / train_aligner) the forward and its backward are libdfx's exact-fp32 training kernels (training.aligner_train_forward);
        otherwise the inference kernels of the latent sampler."""
        if self.cimle and (noise is None or noise.shape[1] != self.noise_dim):
            noise = torch.zeros(x.shape[0], self.noise_dim, device=x.device)             # part_encoders.py:97-98
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if not self.cimle:
                _unsupported("training the part aligner without cimle (pre_norm path)")
            if any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules()) and self.training:
                _unsupported("dropout > 0 inside the part aligner (the shipped configurations use 0)")
            if not x.is_cuda:
                raise RuntimeError("PartAlignerTransformer: CPU not supported")
            from . import training as _training
            if mask is None:
                mask = torch.ones(x.shape[0], self.n_class, device=x.device)
            return _training.aligner_train_forward(dict(self.named_parameters()), x, mask, noise, n_class=self.n_class, zdim=self.zdim,
                                                   n_heads=self.n_heads, d_head=self.d_head, noise_dim=self.noise_dim, noise_scale=self.noise_scale)
        owner = getattr(self, "_owner", None)
        owner = owner() if owner is not None else None
        if owner is None:
            owner = _StandaloneAligner.of(self)
        return owner.sampler().part_aligner(x, mask, noise if self.cimle else None)


class _StandaloneAligner:
    """LatentSampler for a PartAlignerTransformer used on its own (no flows)."""
    _cache = weakref.WeakKeyDictionary()

    def __init__(self, aligner):
        self._al = weakref.ref(aligner)
        self._s = None
        self._ver = None

    @classmethod
    def of(cls, aligner):
        if aligner not in cls._cache:
            cls._cache[aligner] = cls(aligner)
        return cls._cache[aligner]

    def sampler(self):
        al = self._al()
        ver = (_training_generation(list(al.parameters())),) + tuple(p._version for p in al.parameters())
        if self._s is None or ver != self._ver:
            sd = {"part_aligner." + k: v for k, v in al.state_dict().items()}
            self._s = LatentSampler(sd, n_class=al.n_class, zdim=al.zdim, n_heads=al.n_heads, d_head=al.d_head,
                                    cimle=al.cimle, noise_dim=al.noise_dim, noise_scale=al.noise_scale,
                                    device=next(al.parameters()).device)
            self._ver = ver
        return self._s


class PartEncoderForTransformerDecoder(nn.Module):
    def __init__(self, encoder=None, n_class=4, part_aligner=None, fit_loss_weight=1.0, include_z=True,
                 include_part_code=False, include_params=False, use_gt_params=False, encode_ref=False, scale_var=1.0,
                 fit_loss_type=0, origin_scale=False, kl_weight=0.001, use_flow=False, latent_flow_depth=14,
                 latent_flow_hidden_dim=256, gen=False, prior_var=1.0, selective_noise_sampling=False,
                 selective_noise_sampling_global=False, per_part_encoder=False, **kwargs):
        super().__init__()
        if not (gen and include_part_code and include_params) or include_z or encode_ref or per_part_encoder:
            _unsupported("PartEncoder needs gen, include_part_code, include_params and no include_z / encode_ref / per_part_encoder")
        if selective_noise_sampling or selective_noise_sampling_global:
            _unsupported("selective_noise_sampling")
        if use_gt_params and part_aligner is not None:
            _unsupported("use_gt_params together with a part_aligner (stage 1 trains without one, train_chair_stage1.py)")
        if not use_gt_params and part_aligner is None:
            _unsupported("a PartEncoder without part_aligner needs use_gt_params=True")
        for flag in ("gt_param_annealing", "normalize_part_code", "use_gt_params_in_training"):
            if kwargs.get(flag, False):
                _unsupported(flag)
        # training-side settings (part_encoders.py:320-390)
        self.use_gt_params, self.origin_scale, self.kl_weight = use_gt_params, origin_scale, kl_weight
        self.fit_loss_type, self.fit_loss_weight = fit_loss_type, fit_loss_weight
        self.kl_weight_annealing = kwargs.get("kl_weight_annealing", False)
        self.min_kl_weight = kwargs.get("min_kl_weight", 1e-7)
        self.kl_weight_annealing_end_epoch = kwargs.get("kl_weight_annealing_end_epoch", 3000)
        self.detach_params_in_ctx = kwargs.get("detach_params_in_ctx", False)
        self.latent_flow_depth, self.latent_flow_hidden_dim = latent_flow_depth, latent_flow_hidden_dim
        self.encoder_cfg = dict(encoder or {})
        self.zdim = int(self.encoder_cfg.get("zdim", 1024))
        if self.encoder_cfg.get("type", None) == "PointNetV2":   # encode side (get_part_code); not on the generation path
            cfg = {k: v for k, v in self.encoder_cfg.items() if k != "type"}
            self.encoder = PointNetV2(num_anchors=n_class, **cfg)
        self.n_class, self.prior_var, self.gen, self.use_flow = n_class, prior_var, gen, use_flow
        self.log_scale_var = math.log(scale_var)
        if isinstance(part_aligner, dict):
            cfg = dict(part_aligner)
            cfg.pop("type", None)
            part_aligner = PartAlignerTransformer(**cfg)
        self.part_aligner = part_aligner
        if part_aligner is not None:
            self.part_aligner._owner = weakref.ref(self)
        if use_flow:
            self.flow = nn.ModuleList([build_latent_flow(latent_flow_depth, latent_flow_hidden_dim, self.zdim)
                                       for _ in range(n_class)])                          # part_encoders.py:388-390
        self._sampler = None
        self._ver = None

    def sampler(self):
        """The libdfx handle for the current parameters (rebuilt when a parameter was modified in place or reloaded)."""
        own = [p for n, p in self.named_parameters() if not n.startswith("encoder.")]
        ver = (_training_generation(own),) + tuple(p._version for p in own) + (own[0].device,)
        if self._sampler is None or ver != self._ver:
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("encoder.")}
            al = self.part_aligner
            self._sampler = LatentSampler(sd, n_class=self.n_class, zdim=self.zdim, n_heads=al.n_heads, d_head=al.d_head,
                                          cimle=al.cimle, noise_dim=al.noise_dim, noise_scale=al.noise_scale,
                                          prior_var=self.prior_var, log_scale_var=self.log_scale_var,
                                          device=next(self.parameters()).device)
            self._ver = ver
        return self._sampler

    def get_part_code(self, input, seg_flag):
        """part_encoders.py:429-445 (per_part_encoder=False): (means, logvars) of the part codes, each (B, n_class, zdim)."""
        return self.encoder(input, seg_flag)

    def get_params_from_part_code(self, part_code, valid_id, noise=None, gt_mean=None, gt_var=None, **kwargs):
        if self.use_gt_params:                                                            # part_encoders.py:456-458
            return gt_mean, torch.log(gt_var)
        return self.part_aligner(part_code, valid_id, noise=noise)                        # part_encoders.py:447-455

    def gather_all(self, anchor_assignments, anchors=None, variances=None, valid_id=None):
        """part_encoders.py:417-428: per-point (B,3,N) anchors / variances and (B,1,N) flags by part id (index plumbing)."""
        B, N = anchor_assignments.shape
        idx = anchor_assignments.long()[:, None, :]
        dev = anchor_assignments.device
        a = torch.gather(anchors, 2, idx.expand(-1, 3, -1)) if anchors is not None else torch.zeros(B, 3, N, device=dev)
        v = torch.gather(variances, 2, idx.expand(-1, 3, -1)) if variances is not None else torch.zeros(B, 3, N, device=dev)
        f = torch.gather(valid_id[:, None, :].to(torch.float32), 2, idx) if valid_id is not None else torch.ones(B, 1, N, device=dev)
        return a, v, f

    def prepare_ctx(self, part_code, mean, logvar, **kwargs):
        params = torch.cat([mean, torch.exp(logvar + self.log_scale_var)], dim=1)          # part_encoders.py:1317-1326
        return [part_code, params.detach() if self.detach_params_in_ctx else params]

    def get_prior_loss(self, part_code, mean, logvar, valid_id, epoch=-1):
        """part_encoders.py:1143-1182 (use_flow): {'prior_loss', 'kl_weight', per-part log p / entropy / mean / logvar averages}.
        The loss (flows forward + log-det, log-likelihood, entropy) and its backward are libdfx kernels (training.prior_loss)."""
        if not self.use_flow:
            _unsupported("get_prior_loss without use_flow")
        from . import training as _training
        if self.kl_weight_annealing and self.kl_weight_annealing_end_epoch > epoch:
            kl = self.min_kl_weight + (self.kl_weight - self.min_kl_weight) * epoch / self.kl_weight_annealing_end_epoch
        else:
            kl = self.kl_weight
        params = {n: p for n, p in self.named_parameters() if n.startswith("flow.")}
        # `prior_loss_stream` (set by training.stage1_losses for the duration of its call, else None): the whole branch,
        # logging values included, runs on that stream and the caller waits for it before it reads the dict
        side = getattr(self, "prior_loss_stream", None)
        loss, log_p, ent = _training.prior_loss(params, part_code, logvar, valid_id, depth=self.latent_flow_depth,
                                                hidden=self.latent_flow_hidden_dim, prior_var=self.prior_var, kl_weight=kl, stream=side)
        d = {"prior_loss": loss, "kl_weight": torch.ones(1, device=part_code.device) * kl}
        main = torch.cuda.current_stream(part_code.device)
        with torch.no_grad(), torch.cuda.stream(side if side is not None else main):      # logging values only
            nv = valid_id.sum(0)
            mlp, ment = (log_p * valid_id).sum(0) / nv, (ent * valid_id).sum(0) / nv
            mmean, mlv = mean.mean(2).sum(0) / nv, logvar.mean(2).sum(0) / nv
            if side is not None:
                mean.record_stream(side)
                for t in (mlp, ment, mmean, mlv):
                    t.record_stream(main)
        for i in range(self.n_class):
            d[f"log_p_part_{i}"], d[f"entropy_{i}"] = mlp[i], ment[i]
            d[f"part_{i}_mean"], d[f"part_{i}_logvar"] = mmean[i], mlv[i]
        return d

    def get_fit_loss(self, mean, logvar, valid_id, gt_shift, gt_var):
        """part_encoders.py:489-521 for the shipped ``fit_loss_type`` 4 (:514-519; 1 = the same on exp(logvar) / gt_var, :495-500):
        per-shape MSE between the aligner's (mean, logvar) and the ground-truth part parameters over the present parts.  (B,)
        — 96 numbers per shape: host-side elementwise glue like ``gather_all``, nothing for a kernel to win.  Without a
        part_aligner the reference returns zeros(1) (:520-521)."""
        if self.part_aligner is None:
            return torch.zeros(1, device=valid_id.device)
        if self.fit_loss_type not in (1, 4):
            _unsupported(f"fit_loss_type={self.fit_loss_type} (configs/gen_*.py and train_*_stage2.py use 4)")
        if self.fit_loss_type == 4:
            pred, target = torch.cat([mean, logvar], dim=1), torch.cat([gt_shift, torch.log(gt_var)], dim=1)
        else:
            pred, target = torch.cat([mean, torch.exp(logvar)], dim=1), torch.cat([gt_shift, gt_var], dim=1)
        d = (pred - target) ** 2 * valid_id.unsqueeze(1)
        return d.sum(dim=(-1, -2)) / valid_id.sum(dim=-1)

    def _batch(self, pcds, device, attn_key):
        inp = pcds["input"].to(device)
        valid_id = pcds["present"].to(device).to(torch.float32)
        ref = pcds["ref"].to(device).transpose(1, 2)
        seg_mask = pcds["ref_seg_mask"].to(device).to(torch.int32)
        seg_flag = pcds[attn_key].to(device)
        B = ref.shape[0]
        gt_shift = pcds.get("part_shift", torch.zeros(B, 3, self.n_class)).to(device)
        gt_var = pcds.get("part_scale", torch.ones(B, 3, self.n_class)).to(device)
        if not self.origin_scale:
            gt_var = gt_var ** 2
        return inp, valid_id, ref, seg_mask, seg_flag, gt_shift, gt_var

    def _reparameterize(self, m, lv):
        # reparameterize_gaussian (utils/misc.py:282-285): the reference draws eps on the HOST (torch.randn(size).to(mean)) — kept in
        # eval(), so that one torch.manual_seed gives both implementations the same part codes.  train(): drawn on the device (a
        # pageable host->device copy in the middle of the step drains the stream; dropout makes training draws unmatchable anyway)
        eps = torch.randn(lv.size(), device=lv.device) if self.training else torch.randn(lv.size()).to(m)
        return (m + torch.exp(0.5 * lv) * eps).transpose(1, 2)                             # (B, zdim, n_class)

    def forward(self, pcds, device, noise=None, epoch=-1):
        """The encoder's forward (part_encoders.py:1185-1260): part codes from PointNetV2, reparameterisation, prior loss, part
        parameters, per-point gathers, fit loss, ctx for the denoiser.  Two shipped configurations:

        * stage 1 (``use_gt_params``, no part_aligner; train_chair_stage1.py): ground-truth anchors / variances; train() runs the
          PointNetV2 / prior-loss training kernels;
        * gen / stage 2 (a ``part_aligner``, ``fit_loss_type`` 4; configs/gen_*.py, train_*_stage2.py): (mean, logvar) from the
          native aligner on the sampled part codes, ``noise`` (B, num, noise_dim) rows = B * num like the reference's
          ``repeat_interleave`` (:1215-1218).  With gradients enabled (stage-2 training) the aligner runs its exact-fp32 training
          kernels (training.AlignerTrainFn): fit_loss and the denoiser's d ctx[1] reach the aligner's parameters through torch autograd.

        Returns (ctx, mean_per_point, logvar_per_point, flag_per_point, loss_dict, [part_code, mean, logvar, noise])."""
        inp, valid_id, ref, seg_mask, seg_flag, gt_shift, gt_var = self._batch(pcds, device, "ref_attn_map")
        B = ref.shape[0]
        if noise is None:
            noise = pcds["noise"].to(device).unsqueeze(1)
        if self.use_gt_params and noise.shape[1] != 1:
            _unsupported("more than one noise sample per shape in the stage-1 training forward")
        m, lv = self.get_part_code(inp, seg_flag)
        part_code = self._reparameterize(m, lv)
        loss_dict = dict(self.get_prior_loss(part_code, m, lv, valid_id, epoch=epoch))
        num = noise.shape[1]
        noise = noise.reshape(B * num, -1)
        if num > 1:                                                                        # :1215-1218
            part_code, valid_id, seg_mask, ref, gt_shift, gt_var = (t.repeat_interleave(num, dim=0) for t in
                                                                    (part_code, valid_id, seg_mask, ref, gt_shift, gt_var))
        mean, logvar = self.get_params_from_part_code(part_code, valid_id, gt_mean=gt_shift, gt_var=gt_var, ref=ref, noise=noise)
        mean_pp, logvar_pp, flag_pp = self.gather_all(seg_mask, anchors=mean, variances=logvar, valid_id=valid_id)
        loss_dict["fit_loss"] = self.get_fit_loss(mean, logvar, valid_id, gt_shift, gt_var)
        ctx = self.prepare_ctx(part_code, mean, logvar, anchor_assignments=seg_mask)
        return ctx, mean_pp, logvar_pp + self.log_scale_var, flag_pp, loss_dict, [part_code, mean, logvar, noise]

    @torch.no_grad()
    def sample_noise(self, pcds, device, num):
        """part_encoders.py:388-414 (cIMLE noise selection): ``num`` aligner-noise candidates per shape, the fit loss of each, and
        the arg-min.  Returns (noise (B, num, noise_dim), id (B,))."""
        if self.part_aligner is None:
            _unsupported("sample_noise without a part_aligner")
        inp, valid_id, ref, seg_mask, seg_flag, gt_shift, gt_var = self._batch(pcds, device, "attn_map")
        B = inp.shape[0]
        m, lv = self.get_part_code(inp, seg_flag)
        part_code = self._reparameterize(m, lv) if self.gen else m.transpose(1, 2)
        noise = torch.randn(B * num, self.part_aligner.noise_dim).to(device)               # :406 (host draw, like the reference)
        part_code, valid_id, gt_shift, gt_var = (t.repeat_interleave(num, dim=0) for t in (part_code, valid_id, gt_shift, gt_var))
        mean, logvar = self.get_params_from_part_code(part_code, valid_id, noise=noise, gt_mean=gt_shift, gt_var=gt_var)
        fit = self.get_fit_loss(mean, logvar, valid_id, gt_shift, gt_var)
        return noise.reshape(B, num, -1), fit.reshape(B, num).min(1)[1]

    @torch.no_grad()
    def sample_latents(self, sample_num, sample_points, device, fixed_id=None, valid_id=None, epoch=0, K=None,
                       part_code=None, **kwargs):
        """part_encoders.py:1052-1110.  Returns (ctx, mean_per_point, logvar_per_point, seg_mask, valid_id,
        [part_code, mean, logvar, noise]); rows = sample_num * K."""
        al = self.part_aligner
        w = None
        if part_code is None:
            w = torch.randn(sample_num, self.zdim, self.n_class).to(device)               # :1054 — drawn on the HOST like the reference
            #                                                                               (one torch.manual_seed -> the same latents in both); scaled in-kernel
        if al.cimle:
            K = 10 if K is None else K                                                    # :1062
            noise = torch.randn(sample_num * K, al.noise_dim).to(device)                  # :1065 (host draw as well)
            if al.cimle_start_epoch > epoch:
                noise = torch.zeros_like(noise)
        else:
            K, noise = 1, None
        if valid_id is None:
            valid_id = torch.ones(sample_num, self.n_class, device=device)
        fid = [0] * self.n_class if fixed_id is None else [int(v) for v in (fixed_id.tolist() if torch.is_tensor(fixed_id) else fixed_id)]
        out = self.sampler().sample_latents(w, noise, valid_id, fixed_id=fid, K=K, npoints=sample_points, part_code=part_code)
        ctx = [out["part_code"], out["params"]]                                            # prepare_ctx :1317-1326
        return (ctx, out["mean_per_point"], out["logvar_per_point"], out["seg_mask"], out["valid_id"],
                [out["part_code"], out["mean"], out["logvar"], out["noise"]])


@torch.no_grad()
def generate(encoder, diffusion, sample_num, npoints, valid_id=None, fixed_id=None, K=10, epoch=0, seed=None,
             ret_traj=False, ret_interval=20, generator=None):
    """anchor_gen.py:1034-1084 (gen branch) without the batch bookkeeping: latents once per batch, then the fused reverse chain.
    Returns decode's dict + 'pred_seg_mask', 'anchors' (rows, npoints, 3), 'present'.  ``seed=None``: fresh noise per call
    (engine.resolve_seed).  The reference's full output dict is ``networks.AnchorDiffAE.forward``."""
    device = next(encoder.parameters()).device
    ctx, mean_pp, logvar_pp, seg, valid, latents = encoder.sample_latents(sample_num, npoints, device, fixed_id=fixed_id,
                                                                          valid_id=valid_id, epoch=epoch, K=K)
    pred = decode(diffusion, ctx, seg, valid_id=valid, ret_traj=ret_traj, ret_interval=ret_interval, seed=seed, generator=generator)
    pred["pred_seg_mask"] = seg
    pred["anchors"] = mean_pp.transpose(1, 2)
    pred["present"] = valid
    return pred


def attach(ref_encoder):
    """Accelerate a BUILT reference ``PartEncoderForTransformerDecoder`` in place: its ``sample_latents`` is rebound to
    the libdfx-backed one (parameters are read from the instance's own ``state_dict``, so checkpoints loaded before or
    after keep working); the encode side (``forward``, PointNetV2) is left untouched.  Returns the mirror module."""
    al = ref_encoder.part_aligner
    mirror = PartEncoderForTransformerDecoder(
        encoder=dict(zdim=ref_encoder.zdim), n_class=ref_encoder.n_class,
        part_aligner=PartAlignerTransformer(
            in_channels=ref_encoder.zdim, n_heads=al.transformer_blocks[0].attn2.heads,
            d_head=al.inner_dim // al.transformer_blocks[0].attn2.heads, out_channels=al.proj_out.out_features,
            depth=len(al.transformer_blocks), use_linear=al.use_linear, n_class=al.n_class,
            single_attn=al.transformer_blocks[0].single_attn, class_cond=al.class_cond,
            mask_out_unreferenced_code=al.mask_out_unreferenced_code, cimle=al.cimle, noise_dim=al.noise_dim,
            noise_scale=al.noise_scale, cimle_start_epoch=al.cimle_start_epoch, add_class_cond=al.add_class_cond,
            cond_noise_type=al.cond_noise_type),
        include_z=ref_encoder.include_z, include_part_code=ref_encoder.include_part_code,
        include_params=ref_encoder.include_params, use_gt_params=ref_encoder.use_gt_params,
        encode_ref=ref_encoder.encode_ref, scale_var=math.exp(ref_encoder.log_scale_var), gen=ref_encoder.gen,
        fit_loss_type=ref_encoder.fit_loss_type, fit_loss_weight=ref_encoder.fit_loss_weight,
        use_flow=getattr(ref_encoder, "use_flow", False),
        latent_flow_depth=len(ref_encoder.flow[0].chain) if getattr(ref_encoder, "use_flow", False) else 0,
        latent_flow_hidden_dim=ref_encoder.flow[0].chain[0].net_s_t[0].out_features if getattr(ref_encoder, "use_flow", False) else 0,
        prior_var=ref_encoder.prior_var, selective_noise_sampling=ref_encoder.selective_noise_sampling,
        selective_noise_sampling_global=ref_encoder.selective_noise_sampling_global,
        per_part_encoder=ref_encoder.per_part_encoder)
    own = {k for k in mirror.state_dict() if not k.startswith("encoder.")}
    src = weakref.ref(ref_encoder)

    def sync():
        sd = {k: v for k, v in src().state_dict().items() if k in own}
        mirror.to(next(iter(sd.values())).device)
        mirror.load_state_dict(sd, strict=False)

    def sample_latents(*a, **k):
        ver = tuple(p._version for p in src().parameters())
        if getattr(mirror, "_src_ver", None) != ver:
            sync()
            mirror._src_ver = ver
        return mirror.sample_latents(*a, **k)

    ref_encoder.sample_latents = sample_latents
    return mirror

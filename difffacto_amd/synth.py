"""Deterministic synthetic weights / inputs for tests and benchmarks (numpy PCG64).

There is no network and no ``pretrained/*.pth`` in the image, so every measurement runs on
random-init weights of the reference architecture.  Both sides of every parity test (the
reference model in the dev container, the oracle, the HIP path) load the SAME arrays produced
here, keyed by the reference ``state_dict`` names relative to ``diffusion.model.``
(dumped from the built reference model).
"""
import numpy as np

F32 = np.float32

INNER = 128
HEADS = 8
D_HEAD = 16
N_CLASS = 4
ZDIM = 256
CTX_DIM = ZDIM + 6 + N_CLASS + 256  # 522: [part_code | mean | var | eye | t_embed]
IN_CH = 3 + 6 + N_CLASS             # 13:  [x_t | anchors | variances | onehot(seg)]
FF_INNER = 4 * INNER                # 512 (GEGLU projects to 2*512)


def denoiser_param_shapes(depth=5):
    """(name, shape) in reference ``state_dict`` order (TransformerNet, attention.py:318-383)."""
    s = [
        ("pre_norm.weight", (INNER,)), ("pre_norm.bias", (INNER,)),
        ("post_norm.weight", (INNER,)), ("post_norm.bias", (INNER,)),
        ("proj_in.weight", (INNER, IN_CH)), ("proj_in.bias", (INNER,)),
        ("time_embed.net.0.proj.weight", (2048, 256)), ("time_embed.net.0.proj.bias", (2048,)),
        ("time_embed.net.2.weight", (256, 1024)), ("time_embed.net.2.bias", (256,)),
    ]
    for i in range(depth):
        p = f"transformer_blocks.{i}."
        s += [
            (p + "ff.net.0.proj.weight", (2 * FF_INNER, INNER)), (p + "ff.net.0.proj.bias", (2 * FF_INNER,)),
            (p + "ff.net.2.weight", (INNER, FF_INNER)), (p + "ff.net.2.bias", (INNER,)),
            (p + "attn2.to_q.weight", (INNER, INNER)),
            (p + "attn2.to_k.weight", (INNER, CTX_DIM)),
            (p + "attn2.to_v.weight", (INNER, CTX_DIM)),
            (p + "attn2.to_out.0.weight", (INNER, INNER)), (p + "attn2.to_out.0.bias", (INNER,)),
            (p + "norm2.weight", (INNER,)), (p + "norm2.bias", (INNER,)),
            (p + "norm3.weight", (INNER,)), (p + "norm3.bias", (INNER,)),
        ]
    s += [("proj_out.weight", (3, INNER)), ("proj_out.bias", (3,))]
    return s


def make_denoiser_weights(seed=0, depth=5):
    """PyTorch-default-like scale: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear weight and
    bias; LayerNorm weight 1 + U(-.1,.1), bias U(-.1,.1) so the affine terms are exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = {}
    fan = {}
    for name, shape in denoiser_param_shapes(depth):
        if "norm" in name:
            u = rng.uniform(-0.1, 0.1, size=shape)
            W[name] = (u + (1.0 if name.endswith("weight") else 0.0)).astype(F32)
        elif name.endswith("weight"):
            bound = 1.0 / np.sqrt(shape[1])
            fan[name[:-len("weight")]] = bound
            W[name] = rng.uniform(-bound, bound, size=shape).astype(F32)
        else:
            bound = fan[name[:-len("bias")]]
            W[name] = rng.uniform(-bound, bound, size=shape).astype(F32)
    return W


def latent_param_shapes(n_class=N_CLASS, flow_depth=14, flow_hidden=256, depth=5, heads=8, d_head=32, noise_dim=32):
    """(name, shape) of the gen-path encoder parameters, reference ``state_dict`` names relative to
    ``encoder.`` (flow.py:9-19, part_encoders.py:52-86 with configs/gen_chair.py:14-38)."""
    s = []
    half = ZDIM - ZDIM // 2
    for i in range(n_class):
        for l in range(flow_depth):
            p = f"flow.{i}.chain.{l}.net_s_t."
            s += [(p + "0.weight", (flow_hidden, half)), (p + "0.bias", (flow_hidden,)),
                  (p + "2.weight", (flow_hidden, flow_hidden)), (p + "2.bias", (flow_hidden,)),
                  (p + "4.weight", ((ZDIM - half) * 2, flow_hidden)), (p + "4.bias", ((ZDIM - half) * 2,))]
    inner = heads * d_head
    P = "part_aligner."
    s += [(P + "class_emb.weight", (n_class, inner)),
          (P + "pre_norm.weight", (inner,)), (P + "pre_norm.bias", (inner,)),
          (P + "post_norm.weight", (inner,)), (P + "post_norm.bias", (inner,)),
          (P + "proj_in.weight", (inner, ZDIM + noise_dim)), (P + "proj_in.bias", (inner,))]
    for i in range(depth):
        p = f"{P}transformer_blocks.{i}."
        s += [(p + "ff.net.0.proj.weight", (8 * inner, inner)), (p + "ff.net.0.proj.bias", (8 * inner,)),
              (p + "ff.net.2.weight", (inner, 4 * inner)), (p + "ff.net.2.bias", (inner,)),
              (p + "attn2.to_q.weight", (inner, inner)), (p + "attn2.to_k.weight", (inner, inner)),
              (p + "attn2.to_v.weight", (inner, inner)),
              (p + "attn2.to_out.0.weight", (inner, inner)), (p + "attn2.to_out.0.bias", (inner,)),
              (p + "norm2.weight", (inner,)), (p + "norm2.bias", (inner,)),
              (p + "norm3.weight", (inner,)), (p + "norm3.bias", (inner,))]
    s += [(P + "proj_out.weight", (6, inner)), (P + "proj_out.bias", (6,))]
    return s


def make_latent_weights(seed=0, **kw):
    """Same init rule as ``make_denoiser_weights``; class_emb ~ N(0,1) like nn.Embedding."""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    W = {}
    fan = {}
    for name, shape in latent_param_shapes(**kw):
        if "class_emb" in name:
            W[name] = rng.standard_normal(shape).astype(F32)
        elif "norm" in name:
            u = rng.uniform(-0.1, 0.1, size=shape)
            W[name] = (u + (1.0 if name.endswith("weight") else 0.0)).astype(F32)
        elif name.endswith("weight"):
            bound = 1.0 / np.sqrt(shape[1])
            fan[name[:-len("weight")]] = bound
            W[name] = rng.uniform(-bound, bound, size=shape).astype(F32)
        else:
            bound = fan[name[:-len("bias")]]
            W[name] = rng.uniform(-bound, bound, size=shape).astype(F32)
    return W


def pointnet_v2_param_shapes(zdim=ZDIM, num_anchors=N_CLASS):
    """(name, shape) of ``PointNetV2(per_part_mlp=True)`` (python/difffacto/models/encoders/pointnet.py:124-185), BN buffers included."""
    s = []
    for i, (cin, cout) in enumerate(((3, 128), (128, 128), (128, 256), (256, 512)), 1):
        s += [(f"conv{i}.weight", (cout, cin, 1)), (f"conv{i}.bias", (cout,))]
    for i, c in enumerate((128, 128, 256, 512), 1):
        s += [(f"bn{i}.weight", (c,)), (f"bn{i}.bias", (c,)), (f"bn{i}.running_mean", (c,)), (f"bn{i}.running_var", (c,))]
    A = num_anchors
    for name in ("mlp_m", "mlp_v"):
        for idx, (cin, cout) in ((0, (512, 256)), (3, (256, 128)), (6, (128, zdim))):
            s += [(f"{name}.{idx}.weight", (cout * A, cin, 1)), (f"{name}.{idx}.bias", (cout * A,))]
            if idx < 6:
                s += [(f"{name}.{idx + 1}.{k}", (cout * A,)) for k in ("weight", "bias", "running_mean", "running_var")]
    return s


def make_pointnet_v2_weights(seed=0, **kw):
    """Conv weights U(+-1/sqrt(fan_in)); biases / BN shifts / running means U(+-0.2); BN scales U(.8,1.2); running vars U(.5,1.5)."""
    rng = np.random.Generator(np.random.PCG64(seed + 177))
    W = {}
    for name, shape in pointnet_v2_param_shapes(**kw):
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, size=shape)
        elif name.endswith("running_mean") or name.endswith("bias"):
            a = rng.uniform(-0.2, 0.2, size=shape)
        elif len(shape) == 1:
            a = rng.uniform(0.8, 1.2, size=shape)
        else:
            a = rng.uniform(-1, 1, size=shape) / np.sqrt(shape[1])
        W[name] = a.astype(F32)
    return W


def chair_part_distribution():
    """Presence patterns of the 4 chair parts (back, seat, leg, arm): synthetic stand-in for
    ``shapenet_chair_part_distribution`` (datasets/dataset_utils.py:170-179); the data set is
    not in the image, so these are the plausible patterns with fixed weights."""
    pats = np.array([[1, 1, 1, 1], [1, 1, 1, 0], [0, 1, 1, 0], [1, 1, 0, 0], [0, 1, 1, 1]], dtype=F32)
    prob = np.array([0.45, 0.40, 0.05, 0.05, 0.05])
    return pats, prob


def make_latents(B, seed=1, all_valid=False):
    """Synthetic per-shape latents with the shapes/scales ``PartEncoder.sample_latents``
    (part_encoders.py:1052-1110) hands to ``decode``: part_code (B,256,4), mean (B,3,4),
    logvar (B,3,4), valid_id (B,4)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    part_code = rng.standard_normal((B, ZDIM, N_CLASS)).astype(F32)
    mean = (0.3 * rng.standard_normal((B, 3, N_CLASS))).astype(F32)
    logvar = (-3.0 + 0.5 * rng.standard_normal((B, 3, N_CLASS))).astype(F32)
    if all_valid:
        valid = np.ones((B, N_CLASS), dtype=F32)
    else:
        pats, prob = chair_part_distribution()
        valid = pats[rng.choice(len(pats), size=B, p=prob)]
    return part_code, mean, logvar, valid


def make_seg_mask(valid, npoints):
    """part_encoders.py:1105-1106: npoints//n_class consecutive points per part; points of an
    absent part are re-labelled to the first valid part."""
    valid = np.asarray(valid, dtype=F32)
    B, J = valid.shape
    ids = np.arange(J, dtype=F32)[None] * valid + np.argmax(valid, axis=1)[:, None].astype(F32) * (1 - valid)
    seg = np.repeat(ids.astype(np.int32)[:, :, None], npoints // J, axis=2).reshape(B, -1)
    return np.ascontiguousarray(seg)

"""ctypes binding of libdfx.so (the C-ABI declared in include/dfx.h).

Thin by design: pointers are ``tensor.data_ptr()``, the stream is torch's current HIP stream, a
non-zero return code becomes ``RuntimeError`` (the reference's pybind layer raises RuntimeError
through AT_ASSERT, pointnet2_ops/_ext-src/include/utils.h:5-25).

There is NO fallback: if the HIP library is missing or does not load, importing a kernel entry
point raises ``DfxLibraryError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdfx.so")

DFX_PREC_F32 = 0
DFX_PREC_BF16 = 1
DFX_MAX_DEPTH = 8
DFX_ABI_VERSION = 5   # include/dfx.h: the argument lists this binding was written against


class DfxLibraryError(RuntimeError):
    pass


c_fp = ctypes.c_void_p  # device pointers travel as integers


class BlockWeights(ctypes.Structure):
    _fields_ = [(n, c_fp) for n in (
        "norm2_w", "norm2_b", "to_q", "to_k", "to_v", "to_out_w", "to_out_b",
        "norm3_w", "norm3_b", "ff0_w", "ff0_b", "ff2_w", "ff2_b")]


class DenoiserWeights(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int)] + [(n, c_fp) for n in (
        "proj_in_w", "proj_in_b", "pre_norm_w", "pre_norm_b", "post_norm_w", "post_norm_b",
        "proj_out_w", "proj_out_b", "te0_w", "te0_b", "te2_w", "te2_b")] + [("blk", BlockWeights * DFX_MAX_DEPTH)]


class AlignerBlockWeights(ctypes.Structure):
    _fields_ = [(n, c_fp) for n in (
        "norm2_w", "norm2_b", "to_q", "to_k", "to_v", "to_out_w", "to_out_b",
        "norm3_w", "norm3_b", "ff_proj_w", "ff_proj_b", "ff_out_w", "ff_out_b")]


class LatentWeights(ctypes.Structure):
    _fields_ = [("n_class", ctypes.c_int32), ("zdim", ctypes.c_int32), ("flow_depth", ctypes.c_int32),
                ("flow_hidden", ctypes.c_int32), ("flow", ctypes.POINTER(c_fp)),
                ("depth", ctypes.c_int32), ("n_heads", ctypes.c_int32), ("d_head", ctypes.c_int32),
                ("cimle", ctypes.c_int32), ("noise_dim", ctypes.c_int32),
                ("noise_scale", ctypes.c_float), ("prior_var", ctypes.c_float), ("log_scale_var", ctypes.c_float)] + \
               [(n, c_fp) for n in ("proj_in_w", "proj_in_b", "class_emb", "pre_norm_w", "pre_norm_b", "post_norm_w",
                                    "post_norm_b", "proj_out_w", "proj_out_b")] + \
               [("blocks", AlignerBlockWeights * DFX_MAX_DEPTH)]


class PointNetV2Weights(ctypes.Structure):
    _fields_ = [("num_anchors", ctypes.c_int32), ("zdim", ctypes.c_int32), ("reweight_by_anchor", ctypes.c_int32),
                ("bn_eps", ctypes.c_float)] + \
               [(n, c_fp * 4) for n in ("conv_w", "conv_b", "bn_w", "bn_b", "bn_mean", "bn_var")] + \
               [(n, (c_fp * 3) * 2) for n in ("head_w", "head_b")] + \
               [(n, (c_fp * 2) * 2) for n in ("head_bn_w", "head_bn_b", "head_bn_mean", "head_bn_var")]


DFX_MLP_MAX_LAYERS = 4


class SharedMlpTrain(ctypes.Structure):
    """dfx_shared_mlp_train (include/dfx.h): the PointNet++ shared MLP in training mode."""
    _fields_ = [("layers", ctypes.c_int), ("ch", ctypes.c_int * (DFX_MLP_MAX_LAYERS + 1))] + \
               [(n, c_fp * DFX_MLP_MAX_LAYERS) for n in ("conv_w", "conv_b", "bn_w", "bn_b", "bn_mean", "bn_var")] + [("bn_eps", ctypes.c_float), ("relu_mask", ctypes.c_uint32)]


# name -> (restype, argtypes); every symbol include/dfx.h declares
_I, _F, _P, _U64, _SZ, _D = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_double
SIGNATURES = {
    "dfx_version": (_I, []),
    "dfx_abi_version": (_I, []),
    "dfx_last_error": (ctypes.c_char_p, []),
    "dfx_gather_points_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfx_gather_points_grad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfx_furthest_point_sampling_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dfx_fps_max_resident": (_I, []),
    "dfx_ball_query_f32": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "dfx_group_points_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfx_group_points_grad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfx_three_nn_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dfx_three_interpolate_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfx_three_interpolate_grad_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfx_chamfer_forward_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dfx_chamfer_backward_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dfx_denoiser_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(DenoiserWeights), _I, _D, _D, _I, _P]),
    "dfx_denoiser_destroy": (None, [_P]),
    "dfx_denoiser_num_timesteps": (_I, [_P]),
    "dfx_denoiser_precision": (_I, [_P]),
    "dfx_denoiser_w1_fold": (_I, [_P, ctypes.POINTER(ctypes.c_float)]),
    "dfx_debug_w1_fold": (None, [_I]),
    "dfx_debug_w1_fold_channel": (_I, [_P]),
    "dfx_denoiser_get_tables": (_I, [_P, _P]),
    "dfx_shape_ctx_bytes": (_SZ, [_P, _I]),
    "dfx_shape_ctx_prepare": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "dfx_denoise_eps": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "dfx_p_sample": (_I, [_P, _P, _P, _P, _I, _P, _U64, _U64, _P, _P, _I, _I, _P]),
    "dfx_chain_num_snapshots": (_I, [_I, _I]),
    "dfx_sample_chain": (_I, [_P, _P, _P, _P, _P, _U64, _U64, _I, _P, _P, _I, _I, _P]),
    "dfx_latents_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(LatentWeights), _P]),
    "dfx_latents_destroy": (None, [_P]),
    "dfx_flow_reverse": (_I, [_P, _P, _P, _I, _P]),
    "dfx_part_aligner": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "dfx_sample_latents": (_I, [_P, _P, _P, _P, _P, ctypes.POINTER(ctypes.c_int32), _I, _I, _I,
                                _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dfx_aligner_train_workspace_bytes": (_SZ, [_I] * 7),
    "dfx_aligner_train_forward": (_I, [ctypes.POINTER(LatentWeights), _P, _SZ, _P, _P, _P, _P, _P, _I, _P]),
    "dfx_aligner_train_backward": (_I, [ctypes.POINTER(LatentWeights), _P, _SZ, _P, _P, _P, ctypes.POINTER(LatentWeights), _P, _I, _P]),
    "dfx_shared_mlp_create": (_I, [ctypes.POINTER(_P), _I, ctypes.POINTER(ctypes.c_int32)] + [ctypes.POINTER(c_fp)] * 6 + [_F, ctypes.c_uint32, _P]),
    "dfx_pointnet_v2_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(PointNetV2Weights), _P]),
    "dfx_pointnet_v2_destroy": (None, [_P]),
    "dfx_pointnet_v2_forward_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dfx_shared_mlp_destroy": (None, [_P]),
    "dfx_shared_mlp_is_fused": (_I, [_P]),
    "dfx_sa_forward_f32": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfx_fp_forward_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfx_shared_mlp_train_workspace_bytes": (_SZ, [ctypes.POINTER(SharedMlpTrain), _I, _I, _I]),
    "dfx_shared_mlp_train_forward": (_I, [ctypes.POINTER(SharedMlpTrain), _P, _SZ, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "dfx_shared_mlp_train_backward": (_I, [ctypes.POINTER(SharedMlpTrain), _P, _SZ, _P, ctypes.POINTER(SharedMlpTrain), _P, _I, _I, _I, _I, _I, _P]),
    "dfx_emd_workspace_bytes": (_SZ, [_I, _I]),
    "dfx_emd_forward_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "dfx_emd_backward_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dfx_q_sample_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dfx_denoise_eps_t": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "dfx_masked_mse_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dfx_denoiser_train_workspace_bytes": (_SZ, [_I, _I, _I]),
    "dfx_denoiser_train_forward": (_I, [ctypes.POINTER(DenoiserWeights), _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _P]),
    "dfx_denoiser_train_backward": (_I, [ctypes.POINTER(DenoiserWeights), _P, _SZ, _P, ctypes.POINTER(DenoiserWeights), _P, _P, _P, _P, _I, _I, _I, _F, _U64, _P]),
    "dfx_debug_dropout_factors": (_I, [_U64, _I, _F, _P, ctypes.c_longlong, _P]),
    "dfx_debug_train_fused": (None, [_I]),
    "dfx_debug_last_train_path": (ctypes.c_char_p, []),
    "dfx_debug_train_streams": (None, [_I]),
    "dfx_debug_lin_split_k": (None, [_I]),
    "dfx_debug_bn_fused_stats": (None, [_I]),
    "dfx_debug_stats_merge": (None, [_P, _I, _I, _P]),
    "dfx_debug_rowmap": (None, [_I, _P, _P]),
    "dfx_debug_emd_state_global": (None, [_I]),
    "dfx_debug_fps_shape": (None, [_I, _I]),
    "dfx_pointnet_v2_train_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "dfx_pointnet_v2_train_forward": (_I, [ctypes.POINTER(PointNetV2Weights), _P, _SZ, _P, _P, _P, _P, _F, _I, _I, _I, _P]),
    "dfx_pointnet_v2_train_backward": (_I, [ctypes.POINTER(PointNetV2Weights), _P, _SZ, _P, _P, _P, ctypes.POINTER(PointNetV2Weights), _I, _I, _I, _P]),
    "dfx_prior_loss_workspace_bytes": (_SZ, [_I, _I, _I]),
    "dfx_prior_loss_forward": (_I, [ctypes.POINTER(c_fp), _I, _I, _P, _SZ, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P]),
    "dfx_prior_loss_backward": (_I, [ctypes.POINTER(c_fp), _I, _I, _P, _SZ, _P, _F, _F, ctypes.POINTER(c_fp), _P, _P, _I, _P]),
    "dfx_debug_gemm_bf16": (_I, [_I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "dfx_masked_mse_backward_f32": (_I, [_P, _P, _P, _P, _F, _P, _I, _I, _P]),
    "dfx_grad_sumsq_accumulate": (_I, [_P, ctypes.c_longlong, _P, _P, _P]),
    "dfx_adam_step_f32": (_I, [_P, _P, _P, _P, ctypes.c_longlong, _P, _F, _F, _F, _F, _F, _F, _I, _P]),
    "dfx_p_sample_ddim": (_I, [_P, _P, _P, _P, _I, _F, _P, _U64, _U64, _P, _P, _I, _I, _P]),
    "dfx_sample_chain_ddim": (_I, [_P, _P, _P, ctypes.POINTER(ctypes.c_int32), _I, _F, _P, _P, _U64, _U64, _I, _P, _P, _I, _I, _P]),
    "dfx_debug_force_direct": (None, [_I]),
    "dfx_debug_bare_mfma": (_I, [_I, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), _P]),
    "dfx_debug_pipe_waves": (None, [_I]),
    "dfx_debug_trace": (None, [_P, _I]),
    "dfx_set_event_timing": (None, [_I]),
    "dfx_last_kernel_ms": (_F, []),
    "dfx_last_kernel_variant": (ctypes.c_char_p, []),
}

_lib = None


def lib():
    """Load libdfx.so once.  Raises DfxLibraryError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfxLibraryError(
                f"{LIB_PATH} not found: build it with `python -m difffacto_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        # torch ships its own HIP runtime (same SONAME as /opt/rocm's libamdhip64, which libdfx is linked against): load torch's
        # first so that the process holds ONE runtime — in the other order torch ends up on the system copy and finds no device
        import torch  # noqa: F401
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise DfxLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise DfxLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if L.dfx_abi_version() != DFX_ABI_VERSION:
            raise DfxLibraryError(f"{LIB_PATH} has ABI version {L.dfx_abi_version()}, this binding expects {DFX_ABI_VERSION}: "
                                  "rebuild it (python -m difffacto_amd.build --force)")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dfx_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """data_ptr of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())

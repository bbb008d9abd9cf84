"""Test helper: force one chain-kernel variant (include/dfx_debug.h: dfx_debug_pipe_waves) and ASSERT which kernel a launch took
(include/dfx.h: dfx_last_kernel_variant) — a test whose name claims a kernel checks that the kernel ran."""
import contextlib

NAMES = {"bf16": {8: "k_denoise_pipe<8>", 4: "k_denoise_pipe<4>", 2: "k_denoise_pipe<2>", 1: "k_denoise_coop", 64: "k_denoise_pipe2", 16: "k_denoise_coop2", 160: "k_denoise_coop16"},
         "f32": {8: "k_denoise_pipe_f32<8>", 4: "k_denoise_pipe_f32<4>", 2: "k_denoise_pipe_f32<2>", 1: "k_denoise<f32>"}}
AUTO = {"bf16": set(NAMES["bf16"].values()) - {"k_denoise_pipe2"} | {"k_denoise<bf16>"},   # (k_denoise_coop16: the launcher's choice for the smallest batches)
        "f32": set(NAMES["f32"].values())}


@contextlib.contextmanager
def forced(nw):
    """nw = 8 / 4 / 2 wavefronts per workgroup of the pipelined kernel, 1 = co-operative (bf16) or direct (fp32) kernel, 16 = co-operative with two tiles per workgroup, 160 = co-operative on 16-point tiles (161 rules that one out), 64 = pipe2; 0 = automatic."""
    from difffacto_amd import _ffi
    _ffi.lib().dfx_debug_pipe_waves(int(nw))
    try:
        yield
    finally:
        _ffi.lib().dfx_debug_pipe_waves(0)


def ran(prec, nw=0):
    """Assert that the LAST launch took the variant `nw` of precision `prec` (nw = 0: any automatic choice); returns its name."""
    from difffacto_amd.engine import last_kernel_variant
    v = last_kernel_variant()
    if nw:
        assert v == NAMES[prec][nw], f"expected {NAMES[prec][nw]}, the launch took {v}"
    else:
        assert v in AUTO[prec], v
    return v

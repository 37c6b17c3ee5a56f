"""ctypes binding of libclipa_hip.so (the C ABI declared in include/clipa_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel returns an error
code, a RuntimeError is raised (reference-side convention: plain Python exceptions, e.g.
clipa_torch/open_clip/loss.py:40,72).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libclipa_hip.so")

_c = ctypes
_P, _I64, _I32, _F = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float

# name -> (restype, argtypes); must list every symbol of include/clipa_hip.h
SIGNATURES = {
    "clipa_last_error": (_c.c_char_p, []),
    "clipa_version": (_I32, []),
    "clipa_gemm_nt": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _I32, _I32, _P]),
    "clipa_gemm_tn_workspace": (_I64, [_I64, _I64, _I64, _c.POINTER(_I64)]),
    "clipa_gemm_tn": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I32, _P, _I64, _P]),
    "clipa_quantize_rows": (_I32, [_P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P]),
    "clipa_gemm_nt_f8": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _I32, _I32, _I32, _P]),
    "clipa_layernorm_fwd_q8": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "clipa_layernorm_fwd": (_I32, [_P, _P, _P, _P, _I64, _I64, _F, _I32, _I32, _P]),
    "clipa_layernorm_bwd_workspace": (_I64, [_I64, _I64]),
    "clipa_layernorm_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _I32, _I32, _P, _I64, _P]),
    "clipa_layernorm_bwd_y": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _I32, _I32, _P, _I64, _P]),
    "clipa_layernorm_bwd_q8_workspace": (_I64, [_I64, _I64]),
    "clipa_layernorm_bwd_q8": (_I32, [_P] * 13 + [_I64, _I64, _F, _I32, _P, _I64, _P]),
    "clipa_attention_fwd": (_I32, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _P]),
    "clipa_attention_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _P]),
    "clipa_attention_fwd_varlen": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _P]),
    "clipa_attention_bwd_varlen": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64,
                                          _I64, _F, _I32, _P]),
    "clipa_patchify": (_I32, [_P, _P, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _c.POINTER(_F), _c.POINTER(_F), _P]),
    "clipa_resized_crop_workspace": (_I64, [_I64, _I64, _I64]),
    "clipa_resized_crop_u8": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _I64, _P, _P]),
    "clipa_color_jitter_u8": (_I32, [_P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _P]),
    "clipa_assemble_tokens": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "clipa_assemble_tokens_bwd_workspace": (_I64, [_I64, _I64, _I64]),
    "clipa_assemble_tokens_bwd": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _P]),
    "clipa_embed_tokens": (_I32, [_P, _P, _I32, _P, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "clipa_embed_tokens_bwd_workspace": (_I64, [_I64, _I64, _I64, _I64, _I32, _I32]),
    "clipa_embed_tokens_bwd": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P]),
    "clipa_argmax_tokens": (_I32, [_P, _P, _I64, _I64, _P]),
    "clipa_pool_fwd": (_I32, [_P, _P, _P, _I64, _I64, _I64, _I32, _P]),
    "clipa_pool_bwd": (_I32, [_P, _P, _P, _I64, _I64, _I64, _I32, _P]),
    "clipa_gather_rows": (_I32, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "clipa_scatter_rows": (_I32, [_P, _P, _P, _I64, _I64, _I64, _P]),
    "clipa_l2norm_fwd": (_I32, [_P, _P, _P, _P, _I64, _I64, _F, _P]),
    "clipa_l2norm_bwd": (_I32, [_P, _P, _P, _P, _I64, _I64, _P]),
    "clipa_colsum_workspace": (_I64, [_I64, _I64]),
    "clipa_colsum": (_I32, [_P, _P, _I64, _I64, _I64, _P, _I64, _P]),
    "clipa_cast_to_bf16": (_I32, [_P, _I32, _P, _I64, _P]),
    "clipa_cast_bf16_to_f32": (_I32, [_P, _P, _I64, _P]),
    "clipa_transpose_to_bf16": (_I32, [_P, _I32, _P, _I64, _I64, _I64, _I64, _P]),
    "clipa_activation_fwd": (_I32, [_P, _P, _I64, _I32, _P]),
    "clipa_cast_bf16_to_e4m3": (_I32, [_P, _P, _I64, _P]),
    "clipa_cast_e4m3_to_bf16": (_I32, [_P, _P, _I64, _P]),
    "clipa_activation_fwd_e4m3": (_I32, [_P, _P, _I64, _I32, _P]),
    "clipa_quantize_rows_colsum_workspace": (_I64, [_I64, _I64]),
    "clipa_quantize_rows_colsum": (_I32, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P, _I64, _P]),
    "clipa_layernorm_fwd_q8n": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "clipa_gemm_nt_f8q": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I32, _I32, _I32, _P]),
    "clipa_gemm_nt_f8_emit": (_I32, [_P] * 8 + [_I64] * 7 + [_I32, _I32, _P]),
    "clipa_reduce_partial_rows": (_I32, [_P, _P, _I64, _I64, _P]),
    "clipa_rownorm_max": (_I32, [_P, _I64, _I64, _I64, _P, _P]),
    "clipa_absmax_f32": (_I32, [_P, _I64, _P, _P]),
    "clipa_row_bound": (_I32, [_P, _P, _P, _F, _P, _P, _I64, _P]),
    "clipa_rowscale_max": (_I32, [_P, _P, _I64, _P, _P]),
    "clipa_scale_quantize_rows": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P]),
    "clipa_scale_quantize_rows_e4m3": (_I32, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P]),
    "clipa_layernorm_fwd_q8s": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "clipa_gemm_tn_f8_workspace": (_I64, [_I64, _I64, _I64]),
    "clipa_gemm_tn_f8": (_I32, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _F, _P, _I32, _I32, _P, _I64, _P]),
    "clipa_simce_workspace": (_I64, [_I64, _I64]),
    "clipa_simce_fwd": (_I32, [_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _P, _P, _P, _I64, _P]),
    "clipa_simce_bwd": (_I32, [_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _F, _P, _P, _I64, _P, _P, _I64, _P]),
    "clipa_sum_scale": (_I32, [_P, _P, _I64, _F, _I32, _P]),
    "clipa_adamw": (_I32, [_P, _P, _P, _P, _I64, _I32, _I32, _F, _F, _F, _F, _F, _I64, _F, _P]),
    "clipa_adamw_multi": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _F, _F, _F, _F, _I64, _F, _P, _I32, _F, _F, _P]),
    "clipa_grad_sqnorm_multi": (_I32, [_P, _P, _I32, _I32, _P, _P, _I64, _P]),
    "clipa_clip_coef": (_I32, [_P, _F, _P, _P, _P]),
    "clipa_reduce_shards": (_I32, [_P, _P, _I64, _I32, _I32, _I32, _F, _P]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load (once) and return the ctypes handle. Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"clipa_amd: HIP extension {LIB_PATH} is missing - run `python -m clipa_amd.build` "
                "(there is no CPU fallback for the product path)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def debug_set(gemm_nt_variant=0, flags=0):
    """Tests / A-B harnesses only (csrc/internal_hooks.h, not part of the C ABI): force a GEMM kernel family or an ablation
    for this process.  Enables the hook through the environment (CLIPA_DEBUG_HOOKS=1); call with no arguments to reset."""
    lib = load()
    fn = lib.clipa_internal_debug_set
    fn.restype, fn.argtypes = _I32, [_I32, _I32]
    if gemm_nt_variant or flags:
        os.environ["CLIPA_DEBUG_HOOKS"] = "1"
    rc = fn(int(gemm_nt_variant), int(flags))
    if rc != 0:
        raise RuntimeError(f"clipa_internal_debug_set failed (rc={rc}): {last_error()}")


def last_gemm():
    """Kernel family of this process's last GEMM launch (csrc/internal_hooks.h): 1 gemm_nt2, 2 gemm_nta, 3 gemm_tn2,
    4 gemm_tn3, 5 gemm_tna, 6 gemm_f8a, 7 gemm_nt_f8_kernel, 8 gemm_tn8 (four-wave fp8 weight gradient), 9 its byte-gather kernel alone."""
    fn = load().clipa_internal_last_gemm
    fn.restype, fn.argtypes = _I32, []
    return int(fn())


def gemm_counts(reset=False):
    """Launches per GEMM kernel family since the last reset (csrc/internal_hooks.h; tests only): list indexed like last_gemm(),
    [10] / [11] gemm_nta with the e4m3 pre-activation copy / operand epilogue, [12] / [13] the same of gemm_f8a, [14] gemm_f8a with a producer-quantised (e4m3) output, [15] gemm_nta's e4m3-operand GELU-backward that also writes the activation, [0] gemm_f8a's that also emits the weight gradient's activation operand."""
    fn = load().clipa_internal_gemm_counts
    fn.restype, fn.argtypes = _I32, [ctypes.POINTER(ctypes.c_long), _I32, _I32]
    buf = (ctypes.c_long * 16)()
    fn(buf, 16, 1 if reset else 0)
    return list(buf)


def last_error():
    return load().clipa_last_error().decode("utf-8", "replace")


def call(name, *args):
    """Call an int-returning entry point; non-zero return code -> RuntimeError."""
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {last_error()}")


def query(name, *args):
    """Call a value-returning entry point (workspace-size queries)."""
    return getattr(load(), name)(*args)

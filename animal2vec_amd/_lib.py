"""ctypes binding of liba2v_hip.so -- the C-ABI boundary of the MI355X hot path (include/a2v.h).

The library is REQUIRED: there is no CPU or torch fallback anywhere in the product path.  If the
shared object is missing or a symbol cannot be resolved, importing the ops fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# A2V_LIB_PATH: another build of the same library (tools/build_prev.sh: same-box A/B of two revisions); never a fallback
LIB_PATH = os.environ.get("A2V_LIB_PATH") or os.path.join(_HERE, "liba2v_hip.so")

F32, BF16, BF16X3, BF16X3S = 0, 1, 2, 3
EPI_BIAS, EPI_GELU, EPI_RES, EPI_GELU_BWD, EPI_STORE_PRE, EPI_COLSUM, EPI_DROPOUT, EPI_DACT, EPI_X3IMG = 1, 2, 4, 8, 16, 32, 64, 128, 256
EPI_DACT8 = 512
EPI_QKNORM = 1024
ERR_UNSUPPORTED = -4

_p, _i, _ll, _f, _u = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint

# name -> argtypes (mirrors include/a2v.h one to one; tests/test_abi.py cross-checks both)
SIGNATURES = {
    "a2v_conv_gemm": [_i, _p, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _ll, _p, _ll, _i, _p, _p, _ll,
                      _p, _p, _i, _i, _i, _i, _i, _f, _u, _p],
    "a2v_conv_wgrad": [_i, _p, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _i, _p, _ll, _ll, _i, _i, _i,
                       _i, _p],
    "a2v_conv_gemm_rows": [_p, _i, _p, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _ll, _p, _ll, _i, _p, _p, _ll,
                           _p, _p, _i, _i, _i, _i, _i, _f, _u, _p],
    "a2v_conv_wgrad_rows": [_p, _i, _p, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _i, _p, _ll, _ll, _i, _i, _i,
                            _i, _p],
    "a2v_conv_sparse_in": [_i, _p, _ll, _ll, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _ll, _ll, _p, _p, _ll, _i, _i, _p],
    "a2v_conv_wgrad_kept": [_p, _i, _i, _p, _ll, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _ll, _i, _p, _ll, _ll, _i, _i, _i,
                            _i, _p],
    "a2v_linear_wgrad_pair": [_i, _i, _p, _ll, _p, _ll, _p, _ll, _i, _i, _p, _ll, _p, _ll, _p, _ll, _i, _i, _p],
    "a2v_attention_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u, _p],
    "a2v_attention_dropmask_words": [_i, _i, _i],
    "a2v_attention_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u, _p],
    "a2v_attention_bwd_fused_bias": [_i, _i, _i],
    "a2v_layernorm_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _ll, _ll, _f, _i, _f, _u, _p],
    "a2v_layernorm_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _i, _ll, _ll, _i, _f, _u, _i, _p, _p],
    "a2v_instnorm_targets": [_i, C.POINTER(_p), _i, _p, _p, _p, _i, _i, _i, _f, _p],
    "a2v_colsum": [_i, _p, _p, _ll, _i, _ll, _p],
    "a2v_sinc_filters_fwd": [_p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _p],
    "a2v_sinc_filters_bwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _p],
    "a2v_sinc_fused_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "a2v_sinc_fused_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "a2v_sinc_out_channels": [_i, _i],
    "a2v_gather_tokens": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _u, _p],
    "a2v_scatter_tokens": [_i, _p, _p, _p, _i, _i, _i, _i, _p],
    "a2v_clone_reduce": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "a2v_decoder_input": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u, _p],
    "a2v_decoder_input_rng": [_i, _p, _p, _p, _i, _i, _i, _i, _f, _u, _f, _u, _p],
    "a2v_d2v_loss": [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "a2v_ema_update": [_p, _p, _ll, _f, _p],
    "a2v_adam_step": [_p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i, _f, _p, _p, _p, _p],
    "a2v_add_drop": [_i, _p, _p, _p, _p, _ll, _f, _u, _p],
    "a2v_mask_frames": [_i, _p, _p, _p, _p, _i, _i, _i, _f, _u, _i, _p],
    "a2v_sumsq": [_p, _ll, _p, _p],
    "a2v_pack_weight": [_i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "a2v_unpack_wgrad": [_p, _p, _i, _i, _i, _i, _p],
    "a2v_col2im": [_i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _ll, _ll, _p],
    "a2v_cast2d": [_i, _i, _p, _p, _ll, _i, _ll, _ll, _p],
    "a2v_mixup_gain": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "a2v_mixup_apply": [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p],
    "a2v_dropout_mask": [_p, _ll, _f, _u, _p],
    "a2v_dropout_mask2d": [_p, _ll, _i, _f, _u, _p],
    "a2v_debug_set": [_i, _i],
    "a2v_stream_release": [_p],
    "a2v_attention_bwd_direct_dq": [_i, _i, _i],
    "a2v_pack_weight_batch": [_i, _p, _i, _p],
    "a2v_focal_loss": [_i, _p, _p, _p, _p, _ll, _i, _i, _f, _f, _f, _p],
    "a2v_mean_layers": [_i, _p, _i, _p, _ll, _f, _p],
    "a2v_span_mask": [_p, _p, _p, _p, _p, _i, _i, _f, _i, _i, _f, C.c_ulonglong, _p],
    "a2v_maskinfo": [_p, _p, _p, _p, _i, _i, _i, _p],
    "a2v_split_bf16": [_p, _p, _ll, _i, _ll, _ll, _i, _p],
    "a2v_abi_version": [],
    "a2v_query_workspace": [],
    "a2v_bind_workspace": [_p, _p, C.c_size_t],
}
ABI12_ENTRIES = ("a2v_conv_sparse_in", "a2v_conv_wgrad_kept")
# ABI 13: a2v_attention_fwd / a2v_attention_bwd take `dropmask` (one more pointer each, after mdist) + a2v_attention_dropmask_words.
# An ABI-12 library loaded through A2V_LIB_PATH (same-box A/B against the previous round, tools/build_prev.sh) keeps its signatures:
ABI13_ENTRIES = ("a2v_attention_dropmask_words",)
ABI12_SIGNATURES = {
    "a2v_attention_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u, _p],
    "a2v_attention_bwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u, _p],
}
# ABI 14: a2v_attention_fwd takes `qk_norms` (one more pointer, after dropmask); a2v_conv_gemm knows A2V_EPI_QKNORM.
ABI13_SIGNATURES = {
    "a2v_attention_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u, _p],
}
RESTYPES = {"a2v_query_workspace": C.c_size_t}      # everything else returns int

_lib = None


class A2VError(RuntimeError):
    pass


def _load_path(path: str, what: str) -> C.CDLL:
    if not os.path.exists(path):
        raise A2VError(
            f"{path} is missing: the HIP extension{what} is mandatory (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `python animal2vec_amd/build.py`). There is no CPU fallback.")
    # torch first: its wheel carries its own HIP runtime (libamdhip64 under torch/lib).  Loaded before torch, this library pulls in
    # /opt/rocm's copy instead and the process ends up with two runtimes -- torch's owns the devices and streams, ours reports
    # "no device" at the first launch (seen with `python __graft_entry__.py --smoke`, where build() loads the library before anything
    # has imported torch).  With torch already imported the loader resolves the same soname to the one runtime.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    lib.a2v_abi_version.restype = C.c_int
    abi = int(lib.a2v_abi_version())
    lib._a2v_abi = abi
    for name, argtypes in SIGNATURES.items():
        if abi < 13 and os.environ.get("A2V_LIB_PATH"):
            if name in ABI13_ENTRIES:
                continue
            argtypes = ABI12_SIGNATURES.get(name, argtypes)
        elif abi == 13 and os.environ.get("A2V_LIB_PATH"):
            argtypes = ABI13_SIGNATURES.get(name, argtypes)
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            # (an older build loaded through A2V_LIB_PATH for a same-box A/B run may predate the ABI-12 entries: ops.py then takes the
            # calls they replaced; the product library must export everything)
            if name in ABI12_ENTRIES and os.environ.get("A2V_LIB_PATH"):
                continue
            raise A2VError(f"{os.path.basename(path)} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, C.c_int)
    # A2V_DEBUG_SET="key=value,key=value": A/B switches of a2v_debug_set for whole test / bench runs (experiments only).
    # Applied BEFORE the handle is cached: a malformed or unknown entry fails every load(), never half of the switches silently.
    for kv in filter(None, os.environ.get("A2V_DEBUG_SET", "").split(",")):
        try:
            k, v = kv.split("=")
            k, v = int(k), int(v)
        except ValueError as e:
            raise A2VError(f"A2V_DEBUG_SET: malformed entry {kv!r} (want key=value with integers)") from e
        if lib.a2v_debug_set(k, v) != 0:
            raise A2VError(f"A2V_DEBUG_SET: a2v_debug_set({k}, {v}) was refused (unknown key)")
    return lib


def load() -> C.CDLL:
    """Load liba2v_hip.so (never builds, never falls back)."""
    global _lib
    if _lib is None:
        _lib = _load_path(LIB_PATH, "")
    return _lib


_lib_f16 = None
LIB_F16_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liba2v_hip_f16.so")


def load_f16() -> C.CDLL:
    """liba2v_hip_f16.so: the same sources and C ABI with IEEE half as the 16-bit type (csrc/a2v_common.h: A2V_ELT_F16) -- the EMA
    teacher's forward.  Loaded RTLD_LOCAL like the product library, so the two sets of identical symbol names never meet."""
    global _lib_f16
    if _lib_f16 is None:
        _lib_f16 = _load_path(LIB_F16_PATH, " (fp16 teacher build)")
    return _lib_f16


def have_f16() -> bool:
    return os.path.exists(LIB_F16_PATH) and not os.environ.get("A2V_LIB_PATH")


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc <= -1000:
            raise A2VError(f"{what}: HIP launch error {-(rc + 1000)}")
        raise A2VError(f"{what}: {({-1: 'bad argument', -2: 'bad dtype'}).get(rc, rc)}")

"""ctypes binding of the C ABI (include/mila_cdna4.h) for tests and bench plumbing.

PyTorch is used only to own device memory and streams: every compute call goes through
libmila_cdna4.so.  Loading fails loudly when the shared object is missing -- there is no
fallback path of any kind.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmila_cdna4.so")

MILA_OK = 0
MILA_E_INVALID_ARGUMENT = -1
MILA_E_UNSUPPORTED = -2
MILA_E_RUNTIME = -3
MILA_E_SCRATCH_TOO_SMALL = -4

FMT_BF16, FMT_FP8, FMT_FP4 = 0, 1, 2


class MilaError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("mila_cdna4 error %d: %s" % (code, text))
        self.code = code


class InvalidArgument(MilaError, ValueError):
    """reference: std::invalid_argument"""


_lib = None
EXPERIMENTS_PATH = os.path.join(_HERE, "lib", "libmila_cdna4_experiments.so")


class _Libs:
    """the product library, with libmila_cdna4_experiments.so (csrc/experiments/: decode chain + engine, tests / tools only) behind it: a symbol the
    product does not export is looked up there, and only then is that library loaded"""

    def __init__(self, main):
        self.__dict__["_main"] = main
        self.__dict__["_exp"] = None

    def __getattr__(self, name):
        try:
            return getattr(self._main, name)
        except AttributeError:
            if not name.startswith("mila_cdna4_"):
                raise
        if self._exp is None:
            if not os.path.exists(EXPERIMENTS_PATH):
                raise AttributeError("%s: not exported by libmila_cdna4.so, and %s is missing" % (name, EXPERIMENTS_PATH))
            exp = C.CDLL(EXPERIMENTS_PATH)
            exp.mila_cdna4_decode_chain_scratch_bytes.restype = C.c_size_t
            exp.mila_cdna4_decode_engine_scratch_bytes.restype = C.c_size_t
            self.__dict__["_exp"] = exp
        return getattr(self._exp, name)


def load():
    """Load libmila_cdna4.so (import torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -m mila_amd.build` (hipcc --offload-arch=gfx950); "
                          "there is no fallback path" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first so there is one runtime in-process)
    except Exception:
        pass
    main = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)      # global: the experiments library resolves the shared runtime helpers against it
    main.mila_cdna4_last_error.restype = C.c_char_p
    main.mila_cdna4_attn_decode_scratch_bytes.restype = C.c_size_t
    main.mila_cdna4_gemm_staging_bytes.restype = C.c_size_t
    main.mila_cdna4_gemm_workspace_bytes.restype = C.c_size_t
    main.mila_cdna4_gemm_fp8_workspace_bytes.restype = C.c_size_t
    main.mila_cdna4_sample_scratch_bytes.restype = C.c_size_t
    main.mila_cdna4_gemm_w4a8_scratch_bytes.restype = C.c_size_t
    main.mila_cdna4_gemm_w8a8_scratch_bytes.restype = C.c_size_t
    main.mila_cdna4_sample_stochastic_scratch_bytes.restype = C.c_size_t
    main.mila_cdna4_attn_decode_ticket_count.restype = C.c_size_t
    main.mila_cdna4_mha_decode_scratch_bytes.restype = C.c_size_t
    _lib = _Libs(main)
    return _lib


class fused_matvec_args(C.Structure):
    _fields_ = [("y", C.c_void_p), ("x", C.c_void_p), ("W", C.c_void_p), ("scales", C.c_void_p),
                ("norm_w", C.c_void_p), ("post_w", C.c_void_p), ("res", C.c_void_p), ("res_out", C.c_void_p),
                ("post_scale", C.c_float), ("eps", C.c_float), ("fmt", C.c_int), ("K", C.c_int),
                ("N", C.c_int), ("group", C.c_int), ("geglu", C.c_int), ("f32_out", C.c_int),
                ("argmax_scratch", C.c_void_p), ("argmax_scratch_bytes", C.c_size_t), ("argmax_blocks", C.POINTER(C.c_int))]


class decode_chain_args(C.Structure):
    _fields_ = [("attn", C.c_void_p), ("res", C.c_void_p), ("res_out", C.c_void_p), ("y", C.c_void_p),
                ("W_o", C.c_void_p), ("s_o", C.c_void_p), ("W_gate_up", C.c_void_p), ("s_gate_up", C.c_void_p),
                ("W_down", C.c_void_p), ("s_down", C.c_void_p), ("W_next", C.c_void_p), ("s_next", C.c_void_p),
                ("post_attn_w", C.c_void_p), ("pre_ffn_w", C.c_void_p), ("post_ffn_w", C.c_void_p),
                ("next_norm_w", C.c_void_p), ("layer_scalar", C.c_float), ("eps", C.c_float),
                ("fmt", C.c_int), ("group", C.c_int), ("next_fmt", C.c_int), ("next_group", C.c_int),
                ("f32_out", C.c_int), ("D", C.c_int), ("F", C.c_int), ("K_attn", C.c_int), ("N_next", C.c_int),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t)]


def _ptr(t):
    """device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "the C ABI takes dense tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tune(name, value):
    """set a named tuning variable (csrc/internal.h; the process must have set MILA_CDNA4_TUNING=1 before the library loaded)"""
    check(load().mila_cdna4_tune(name.encode(), C.c_int(int(value))))


def tune_reset():
    check(load().mila_cdna4_tune_reset())


def last_form():
    """the kernel forms this thread's Linear / attention entry points ran since the previous call, as a list; clears the record"""
    lib = load()
    lib.mila_cdna4_last_form.restype = C.c_size_t
    buf = C.create_string_buffer(512)
    lib.mila_cdna4_last_form(buf, C.c_size_t(512))
    return [f for f in buf.value.decode().split("+") if f]


def check(rc):
    if rc == MILA_OK:
        return
    text = load().mila_cdna4_last_error().decode()
    if rc == MILA_E_INVALID_ARGUMENT:
        raise InvalidArgument(rc, text)
    raise MilaError(rc, text)


def call(name, *args):
    """call mila_cdna4_<name>(*args, current torch stream) and raise on a non-zero status."""
    fn = getattr(load(), "mila_cdna4_" + name)
    conv = []
    for a in args:
        if a is None or isinstance(a, (C.c_void_p, C.c_float, C.c_int, C.c_int64, C.c_size_t)):
            conv.append(a)
        elif isinstance(a, float):
            conv.append(C.c_float(a))
        elif isinstance(a, int):
            conv.append(C.c_int(a))
        elif hasattr(a, "data_ptr"):
            conv.append(_ptr(a))
        else:
            conv.append(a)
    check(fn(*conv, _stream()))


EXPORTED = [
    "last_error", "abi_version", "device_count", "set_device", "device_info", "stream_create",
    "stream_destroy", "stream_synchronize", "malloc", "free", "host_alloc_pinned", "host_free_pinned",
    "memcpy_h2d", "memcpy_d2h", "memcpy_d2d", "memset_zero",
    "matvec_bf16", "matvec_bf16_qfp8", "matvec_bf16_qfp4", "matvec_f32out",
    "gemm_bf16", "gemm_gelu_bf16", "gemm_workspace_bytes", "gemm_bf16_ws", "gemm_bf16_w8a16", "gemm_bf16_w4a16", "gemm_staging_bytes", "gemm_bf16_w8a16_staged", "gemm_bf16_w4a16_staged",
    "fp4_weight_fp8_scale", "upcast_fp4_to_fp8", "quantize_fp8_per_token", "gemm_fp8_applicable", "gemm_fp8_scaled", "gemm_fp8_workspace_bytes", "gemm_fp8_scaled_ws",
    "gemm_w4a8_scratch_bytes", "gemm_bf16_w4a8", "gemm_geglu_w4a8_applicable", "gemm_geglu_bf16_w4a8",
    "gemm_geglu_applicable", "gemm_geglu_preferred", "gemm_geglu_bf16", "gemm_geglu_bf16_w8a16_staged", "gemm_geglu_bf16_w4a16_staged",
    "quantize_fp8_per_channel", "quantize_fp4_per_group",
    "kv_write_bf16", "attn_decode_scratch_bytes", "attn_decode_bf16", "attn_prefill_bf16", "mha_bf16", "mha_kv_write_bf16", "mha_decode_scratch_bytes", "mha_decode_bf16",
    "rmsnorm_bf16", "rmsnorm_fp32", "layernorm_bf16", "layernorm_fp32", "softmax_fp32", "softmax_bf16",
    "gelu_bf16", "gelu_fp32", "geglu_bf16", "residual_bf16", "residual_fp32",
    "rope_build_cache", "rope_forward_bf16",
    "embedding_gather_bf16", "embedding_gather_bf16_qfp8", "lpe_bf16", "split3_bf16", "scale_bf16",
    "convert_f32_to_bf16", "convert_bf16_to_f32", "fill_uniform_bf16",
    "sample_scratch_bytes", "sample_argmax_fp32", "sample_argmax_bf16",
    "sample_stochastic_scratch_bytes", "sample_stochastic_fp32", "sample_stochastic_bf16",
    "fused_norm_matvec", "fused_qkv_post", "fused_qkv_post_prefill", "fused_tail_norm_bf16", "fused_tail_norm_quant_bf16",
    "attn_decode_bf16_devpos", "fused_qkv_post_devpos", "advance_position", "advance_position_snapshot", "snapshot_token", "sample_argmax_advance_fp32", "sample_argmax_final_advance", "fused_attn_decode_batch_bf16", "fused_attn_decode_bf16",
    "dequantize_to_bf16", "gemm_geglu_fp8_scaled",
    "gemm_fp8_w8a8_ws", "gemm_geglu_fp8_w8a8", "gemm_w8a8_scratch_bytes", "gemm_bf16_w8a8", "gemm_geglu_bf16_w8a8",
    "matvec_fp32", "gemm_fp32", "mha_fp32", "mha_kv_write_fp32", "mha_decode_fp32", "lpe_fp32", "rope_forward_fp32",
]

# csrc/internal.h: test / tuning hooks and the measured-slower experiments -- exported, but not part of the drop-in ABI
INTERNAL = [
    "tune", "tune_get", "tune_reset", "tune_list", "last_form",
    "decode_engine_debug",
    "selftest_decode", "selftest_wave_reduce", "selftest_mfma_fp8", "stream_copy", "stream_read",
    "attn_decode_split_count", "fused_attn_decode_partials_bf16", "matvec_attn_combine",
    "decode_chain_scratch_bytes", "decode_chain_init", "decode_chain_status", "decode_chain",
    "decode_engine_scratch_bytes", "decode_engine_init", "decode_engine_status", "decode_engine_applicable", "decode_engine",
    "attn_decode_ticket_count", "fused_attn_decode_onepass_bf16", "prefetch_l3", "fused_attn_decode_ex",
    "exp_gemm4w_bf16", "exp_gemm4w_geglu_bf16",
]
# ... of which these live in libmila_cdna4_experiments.so
EXPERIMENTS_LIB = [
    "decode_chain_scratch_bytes", "decode_chain_init", "decode_chain_status", "decode_chain",
    "decode_engine_scratch_bytes", "decode_engine_init", "decode_engine_status", "decode_engine_applicable", "decode_engine", "decode_engine_debug",
    "exp_gemm4w_bf16", "exp_gemm4w_geglu_bf16",
]

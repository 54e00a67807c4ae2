"""ctypes binding of the CPU oracle (oracle/mila_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (mila_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(ORACLE_DIR, "_build", "libmila_oracle.so")
_REF = os.path.join(ORACLE_DIR, "_ref", "libmila_ref_act.so")


def build(force=False):
    src = os.path.join(ORACLE_DIR, "mila_oracle.c")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"])


build()
lib = C.CDLL(_LIB)

f32p = C.POINTER(C.c_float)
u16p = C.POINTER(C.c_uint16)
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
i64 = C.c_int64


def _p(a, typ):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return a.ctypes.data_as(typ)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- scalar helpers -------------------------------------------------------------------------
lib.orc_f32_to_bf16.restype = C.c_uint16
lib.orc_f32_to_bf16.argtypes = [C.c_float]
lib.orc_bf16_to_f32.restype = C.c_float
lib.orc_bf16_to_f32.argtypes = [C.c_uint16]
lib.orc_f32_to_e4m3.restype = C.c_uint8
lib.orc_f32_to_e4m3.argtypes = [C.c_float]
lib.orc_e4m3_to_f32.restype = C.c_float
lib.orc_e4m3_to_f32.argtypes = [C.c_uint8]
lib.orc_f32_to_e2m1.restype = C.c_uint8
lib.orc_f32_to_e2m1.argtypes = [C.c_float]
lib.orc_e2m1_to_f32.restype = C.c_float
lib.orc_e2m1_to_f32.argtypes = [C.c_uint8]
lib.orc_gelu_tanh.restype = C.c_float
lib.orc_gelu_tanh.argtypes = [C.c_float]
lib.orc_silu.restype = C.c_float
lib.orc_silu.argtypes = [C.c_float]
lib.orc_softcap.restype = C.c_float
lib.orc_softcap.argtypes = [C.c_float, C.c_float]
lib.orc_fp8_weight_scale_from_groups.restype = C.c_float
lib.orc_cpu_lpe.restype = C.c_int
lib.orc_embedding_gather.restype = C.c_int


def to_bf16_bits(x):
    """float32 array -> uint16 bf16 bit patterns (RNE)."""
    x = _f(x)
    out = np.empty(x.shape, dtype=np.uint16)
    lib.orc_f32_to_bf16_array(_p(out, u16p), _p(x, f32p), i64(x.size))
    return out


def from_bf16_bits(h):
    h = np.ascontiguousarray(h, dtype=np.uint16)
    out = np.empty(h.shape, dtype=np.float32)
    lib.orc_bf16_to_f32_array(_p(out, f32p), _p(h, u16p), i64(h.size))
    return out


def round_bf16(x):
    """float32 array rounded to bf16-representable float32 values."""
    return from_bf16_bits(to_bf16_bits(x))


E4M3_LUT = np.array([lib.orc_e4m3_to_f32(i) for i in range(256)], dtype=np.float32)
E2M1_LUT = np.array([lib.orc_e2m1_to_f32(i) for i in range(16)], dtype=np.float32)


# ---- reference CPU backend ops ------------------------------------------------------------------
def cpu_linear(X, W, B=None, path="auto"):
    X, W = _f(X), _f(W)
    N, K = W.shape
    batch = X.size // K
    Y = np.empty(X.shape[:-1] + (N,), dtype=np.float32)
    Bp = _p(_f(B), f32p) if B is not None else None
    fn = {"auto": lib.orc_cpu_linear, "naive": lib.orc_cpu_linear_naive,
          "unrolled": lib.orc_cpu_linear_unrolled}[path]
    fn(_p(Y, f32p), _p(X, f32p), _p(W, f32p), Bp, i64(batch), i64(K), i64(N))
    return Y


def cpu_gelu(X):
    X = _f(X)
    Y = np.empty_like(X)
    lib.orc_cpu_gelu(_p(Y, f32p), _p(X, f32p), i64(X.size))
    return Y


def cpu_softmax(X, axis=-1):
    X = _f(X)
    axis = axis % X.ndim
    outer = int(np.prod(X.shape[:axis], dtype=np.int64))
    dim = X.shape[axis]
    inner = int(np.prod(X.shape[axis + 1:], dtype=np.int64))
    Y = np.empty_like(X)
    lib.orc_cpu_softmax(_p(Y, f32p), _p(X, f32p), i64(outer), i64(dim), i64(inner))
    return Y


def cpu_layernorm(X, w, b, eps=1e-5, return_stats=False):
    X = _f(X)
    dim = X.shape[-1]
    outer = X.size // dim
    Y = np.empty_like(X)
    mean = np.empty(outer, dtype=np.float32)
    rstd = np.empty(outer, dtype=np.float32)
    lib.orc_cpu_layernorm(_p(Y, f32p), _p(mean, f32p), _p(rstd, f32p), _p(X, f32p),
                          _p(_f(w), f32p) if w is not None else None,
                          _p(_f(b), f32p) if b is not None else None,
                          i64(outer), i64(dim), i64(1), C.c_float(eps))
    return (Y, mean, rstd) if return_stats else Y


def cpu_residual(A, B):
    A, B = _f(A), _f(B)
    Y = np.empty_like(A)
    lib.orc_cpu_residual(_p(Y, f32p), _p(A, f32p), _p(B, f32p), i64(A.size))
    return Y


def cpu_lpe(tokens, wte, wpe, out_T=None):
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    wte, wpe = _f(wte), _f(wpe)
    B, T = tokens.shape
    Cc = wte.shape[1]
    out_T = out_T or T
    Y = np.zeros((B, out_T, Cc), dtype=np.float32)
    rc = lib.orc_cpu_lpe(_p(Y, f32p), _p(tokens, i32p), _p(wte, f32p), _p(wpe, f32p),
                         i64(B), i64(T), i64(Cc), i64(out_T), i64(wte.shape[0]))
    if rc != 0:
        raise IndexError("token index outside vocabulary range")
    return Y


def cpu_mha(X, NH):
    X = _f(X)
    B, T, C3 = X.shape
    Cc = C3 // 3
    Y = np.empty((B, T, Cc), dtype=np.float32)
    lib.orc_cpu_mha(_p(Y, f32p), _p(X, f32p), C.c_int(B), C.c_int(T), C.c_int(Cc), C.c_int(NH))
    return Y


def cpu_gpt2_forward(tokens, params, C_, L, NH, V, maxT):
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    B, T = tokens.shape
    params = [_f(p) for p in params]
    arr = (f32p * len(params))(*[_p(p, f32p) for p in params])
    logits = np.empty((B, T, V), dtype=np.float32)
    lib.orc_cpu_gpt2_forward(_p(logits, f32p), _p(tokens, i32p), arr, C.c_int(B), C.c_int(T),
                             C.c_int(C_), C.c_int(L), C.c_int(NH), C.c_int(V), C.c_int(maxT))
    return logits


# ---- CUDA-arithmetic restatements ---------------------------------------------------------------
def geglu(X):
    X = _f(X)
    half = X.shape[-1] // 2
    tokens = X.size // (2 * half)
    Y = np.empty(X.shape[:-1] + (half,), dtype=np.float32)
    lib.orc_geglu(_p(Y, f32p), _p(X, f32p), i64(tokens), i64(half))
    return Y


def gelu_tanh(X):
    return cpu_gelu(X)


def rmsnorm(X, w, b=None, eps=1e-5, w_offset=0.0, inner=1, return_rstd=False):
    X = _f(X)
    if inner == 1:
        dim = X.shape[-1]
        outer = X.size // dim
    else:
        dim = X.shape[-2]
        outer = X.size // (dim * inner)
    Y = np.empty_like(X)
    rstd = np.empty(outer * inner, dtype=np.float32)
    lib.orc_rmsnorm(_p(Y, f32p), _p(rstd, f32p), _p(X, f32p),
                    _p(_f(w), f32p) if w is not None else None,
                    _p(_f(b), f32p) if b is not None else None,
                    i64(outer), i64(dim), i64(inner), C.c_float(eps), C.c_float(w_offset))
    return (Y, rstd) if return_rstd else Y


def rope_build_cache(max_seq, head_dim, base, rotary_dim=0):
    half = head_dim // 2
    cos = np.empty((max_seq, half), dtype=np.float32)
    sin = np.empty((max_seq, half), dtype=np.float32)
    lib.orc_rope_build_cache(_p(cos, f32p), _p(sin, f32p), C.c_int(max_seq), C.c_int(head_dim),
                             C.c_float(base), C.c_int(rotary_dim))
    return cos, sin


def rope_rotate(X, cos, sin, pos_offset=0):
    """X [B,T,n_heads,head_dim]."""
    X = _f(X)
    B, T, H, D = X.shape
    Y = np.empty_like(X)
    lib.orc_rope_rotate(_p(Y, f32p), _p(X, f32p), _p(_f(cos), f32p), _p(_f(sin), f32p),
                        i64(B), i64(T), i64(H), i64(D), i64(pos_offset))
    return Y


def quantize_fp8_per_channel(W_bf16_bits):
    W = np.ascontiguousarray(W_bf16_bits, dtype=np.uint16)
    N, K = W.shape
    q = np.empty((N, K), dtype=np.uint8)
    s = np.empty(N, dtype=np.float32)
    lib.orc_quantize_fp8_per_channel(_p(q, u8p), _p(s, f32p), _p(W, u16p), i64(N), i64(K))
    return q, s


def quantize_fp4_per_group(W_bf16_bits, group=128):
    W = np.ascontiguousarray(W_bf16_bits, dtype=np.uint16)
    N, K = W.shape
    q = np.empty((N, K // 2), dtype=np.uint8)
    s = np.empty((N, K // group), dtype=np.float32)
    lib.orc_quantize_fp4_per_group(_p(q, u8p), _p(s, f32p), _p(W, u16p), i64(N), i64(K),
                                   C.c_int(group))
    return q, s


def fp8_weight_scale_from_groups(scales):
    s = _f(scales).reshape(-1)
    return float(lib.orc_fp8_weight_scale_from_groups(_p(s, f32p), i64(s.size)))


def dequant_fp8(q, s):
    N, K = q.shape
    W = np.empty((N, K), dtype=np.float32)
    lib.orc_dequant_fp8(_p(W, f32p), _p(np.ascontiguousarray(q), u8p), _p(_f(s), f32p), i64(N), i64(K))
    return W


def dequant_fp4(q, s, group=128):
    N, K2 = q.shape
    K = K2 * 2
    W = np.empty((N, K), dtype=np.float32)
    lib.orc_dequant_fp4(_p(W, f32p), _p(np.ascontiguousarray(q), u8p), _p(_f(s), f32p), i64(N),
                        i64(K), C.c_int(group))
    return W


def _bias_bits(bias):
    return None if bias is None else np.ascontiguousarray(bias, dtype=np.uint16)


def linear_bf16w(X, W_bits, bias_bits=None):
    X = _f(X)
    W = np.ascontiguousarray(W_bits, dtype=np.uint16)
    N, K = W.shape
    M = X.size // K
    Y = np.empty(X.shape[:-1] + (N,), dtype=np.float32)
    b = _bias_bits(bias_bits)
    lib.orc_linear_bf16w(_p(Y, f32p), _p(X, f32p), _p(W, u16p), _p(b, u16p), i64(M), i64(K), i64(N))
    return Y


def linear_fp8w(X, q, s, bias_bits=None):
    X = _f(X)
    N, K = q.shape
    M = X.size // K
    Y = np.empty(X.shape[:-1] + (N,), dtype=np.float32)
    b = _bias_bits(bias_bits)
    lib.orc_linear_fp8w(_p(Y, f32p), _p(X, f32p), _p(np.ascontiguousarray(q), u8p), _p(_f(s), f32p),
                        _p(b, u16p), i64(M), i64(K), i64(N))
    return Y


def linear_fp4w(X, q, s, group=128, bias_bits=None):
    X = _f(X)
    N, K2 = q.shape
    K = 2 * K2
    M = X.size // K
    Y = np.empty(X.shape[:-1] + (N,), dtype=np.float32)
    b = _bias_bits(bias_bits)
    lib.orc_linear_fp4w(_p(Y, f32p), _p(X, f32p), _p(np.ascontiguousarray(q), u8p), _p(_f(s), f32p),
                        _p(b, u16p), i64(M), i64(K), i64(N), C.c_int(group))
    return Y


def upcast_fp4_to_fp8(packed, scales, weight_fp8_scale, group):
    """e4m3 codes of the W4A8 weight staging (CudaW4A16Gemm.cu:300-323)"""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    N, K = packed.shape[0], packed.shape[1] * 2
    out = np.empty((N, K), dtype=np.uint8)
    lib.orc_upcast_fp4_to_fp8(_p(out, u8p), _p(packed, u8p), _p(_f(scales), f32p), C.c_float(weight_fp8_scale), i64(N), i64(K), C.c_int(group))
    return out


def quantize_act_fp8_per_token(X):
    X = _f(X)
    K = X.shape[-1]
    M = X.size // K
    q = np.empty((M, K), dtype=np.uint8)
    s = np.empty(M, dtype=np.float32)
    lib.orc_quantize_act_fp8_per_token(_p(q, u8p), _p(s, f32p), _p(X, f32p), i64(M), i64(K))
    return q, s


def linear_fp8a_fp8w(Xq, ts, Wq, w_row_scale=None, w_tensor_scale=1.0, bias_bits=None):
    M, K = Xq.shape
    N = Wq.shape[0]
    Y = np.empty((M, N), dtype=np.float32)
    b = _bias_bits(bias_bits)
    lib.orc_linear_fp8a_fp8w(_p(Y, f32p), _p(np.ascontiguousarray(Xq), u8p), _p(_f(ts), f32p),
                             _p(np.ascontiguousarray(Wq), u8p),
                             _p(_f(w_row_scale), f32p) if w_row_scale is not None else None,
                             C.c_float(w_tensor_scale), _p(b, u16p), i64(M), i64(K), i64(N))
    return Y


def gqa_attention(q, k, v, pos_offset=0, window=0, scale=1.0):
    """q [B,Tq,NH,HS]; k,v [B,Tk,NKV,HS] (linear history) -> [B,Tq,NH*HS]."""
    q, k, v = _f(q), _f(k), _f(v)
    B, Tq, NH, HS = q.shape
    Tk, NKV = k.shape[1], k.shape[2]
    out = np.empty((B, Tq, NH * HS), dtype=np.float32)
    lib.orc_gqa_attention(_p(out, f32p), _p(q, f32p), _p(k, f32p), _p(v, f32p), C.c_int(B),
                          C.c_int(Tq), C.c_int(Tk), C.c_int(NH), C.c_int(NKV), C.c_int(HS),
                          C.c_int(pos_offset), C.c_int(window), C.c_float(scale))
    return out


def kv_write(Kc, Vc, k, v, start_pos):
    """Kc,Vc [B,NKV,capacity,HS] float32 updated in place; k,v [B,chunk,NKV,HS]."""
    k, v = _f(k), _f(v)
    B, chunk, NKV, HS = k.shape
    cap = Kc.shape[2]
    lib.orc_kv_write(_p(Kc, f32p), _p(Vc, f32p), _p(k, f32p), _p(v, f32p), C.c_int(B),
                     C.c_int(chunk), C.c_int(NKV), C.c_int(HS), C.c_int(start_pos), C.c_int(cap))


def kv_ring_to_linear(Kc, first, Tk):
    B, NKV, cap, HS = Kc.shape
    out = np.empty((B, Tk, NKV, HS), dtype=np.float32)
    lib.orc_kv_ring_to_linear(_p(out, f32p), _p(_f(Kc), f32p), C.c_int(B), C.c_int(NKV),
                              C.c_int(HS), C.c_int(cap), C.c_int(first), C.c_int(Tk))
    return out


def embedding_gather(tokens, table, scale=0.0):
    tokens = np.ascontiguousarray(tokens, dtype=np.int32).reshape(-1)
    table = _f(table)
    V, Cc = table.shape
    Y = np.empty((tokens.size, Cc), dtype=np.float32)
    rc = lib.orc_embedding_gather(_p(Y, f32p), _p(tokens, i32p), _p(table, f32p), i64(tokens.size),
                                  i64(Cc), i64(V), C.c_float(scale))
    if rc != 0:
        raise IndexError("token index outside vocabulary range")
    return Y


lib.orc_sample_stochastic.restype = C.c_int
lib.orc_sample_stochastic.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]


def sample_stochastic(logits, softcap, temperature, top_k, top_p, r):
    """(token, margins[3]) of the reference's multinomial sampler semantics (oracle/mila_oracle.c orc_sample_stochastic)"""
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    m = np.zeros(3, dtype=np.float64)
    tok = lib.orc_sample_stochastic(lg.ctypes.data, lg.size, softcap, temperature, top_k, top_p, r, m.ctypes.data)
    return tok, m


def softcap(x, cap=30.0):
    return np.array([lib.orc_softcap(float(v), cap) for v in np.asarray(x).reshape(-1)],
                    dtype=np.float32).reshape(np.asarray(x).shape)


# ---- the one reference source that builds here (oracle/_ref) -------------------------------------
def ref_activation_lib():
    """The reference's own activation functors (compiled from /root/reference), or None."""
    if not os.path.exists(_REF):
        return None
    r = C.CDLL(_REF)
    for n in ("ref_gelu_tanh", "ref_silu", "ref_relu", "ref_tanh", "ref_sigmoid", "ref_mish"):
        getattr(r, n).restype = C.c_float
        getattr(r, n).argtypes = [C.c_float]
    return r

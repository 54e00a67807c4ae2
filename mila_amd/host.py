"""ctypes binding of libmila_host.so: the C++ host mirror's model runners (GemmaTransformer<TWeightQuant>
on DeviceType::Rocm).  Used by bench.py and the model-level tests; loading fails loudly if the
library is missing."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmila_host.so")
POLICIES = {"bf16": 0, "fp8": 1, "fp4": 2}
_lib = None


class GemmaConfigC(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("vocab_size", "embedding_dim", "num_layers", "num_heads", "num_kv_heads",
                                         "head_dim", "hidden_dim", "global_head_dim", "num_global_kv_heads", "window",
                                         "sliding_window_pattern", "global_rotary_dim", "bounded_local_kv")]


GEMMA4_12B = dict(vocab_size=262144, embedding_dim=3840, num_layers=48, num_heads=16, num_kv_heads=8, head_dim=256,
                  hidden_dim=15360, global_head_dim=512, num_global_kv_heads=1, window=1024, sliding_window_pattern=6,
                  global_rotary_dim=128)


def load():
    global _lib
    if _lib is None:
        capi.load()      # libmila_cdna4 first (and torch's HIP runtime before both)
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -m mila_amd.build`; there is no fallback path" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.mila_host_last_error.restype = C.c_char_p
        _lib.mila_gemma_create.restype = C.c_void_p
        _lib.mila_gemma_create.argtypes = [C.c_int, C.POINTER(GemmaConfigC), C.c_int64, C.c_int64, C.c_uint64, C.c_int]
        _lib.mila_gemma_destroy.argtypes = [C.c_void_p]
        _lib.mila_gemma_init_synthetic.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.mila_gemma_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        _lib.mila_gemma_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_void_p]
        _lib.mila_gemma_time_decode.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.mila_gemma_time_dominant_kernel.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.mila_gemma_time_prefill.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        _lib.mila_gemma_info.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _lib.mila_gemma_generate.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        _lib.mila_gemma_generate_sampled.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint32, C.c_void_p]
        _lib.mila_gemma_set_fp8_activation_prefill.argtypes = [C.c_void_p, C.c_int]
        _lib.mila_gemma_set_fused_prefill.argtypes = [C.c_void_p, C.c_int]
        _lib.mila_gemma_set_resident_prefill_weights.argtypes = [C.c_void_p, C.c_int]
        _lib.mila_gemma_save_safetensors.argtypes = [C.c_void_p, C.c_char_p]
        _lib.mila_gemma_load_safetensors.argtypes = [C.c_void_p, C.c_char_p]
        _lib.mila_gemma_save_milabin.argtypes = [C.c_void_p, C.c_char_p]
        _lib.mila_gemma_load_pretrained.argtypes = [C.c_void_p, C.c_char_p]
        _lib.mila_pretrained_list.restype = C.c_int64
        _lib.mila_pretrained_list.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
        _lib.mila_pretrained_to_milabin.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        _lib.mila_pretrained_metadata_roundtrip.restype = C.c_int64
        _lib.mila_pretrained_metadata_roundtrip.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
        _lib.mila_safetensors_list.restype = C.c_int64
        _lib.mila_safetensors_list.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
        _lib.mila_safetensors_copy.argtypes = [C.c_char_p, C.c_char_p]
        _lib.mila_gpt_last_error.restype = C.c_char_p
        _lib.mila_gpt_create.restype = C.c_void_p
        _lib.mila_gpt_create.argtypes = [C.c_int64] * 7
        _lib.mila_gpt_destroy.argtypes = [C.c_void_p]
        _lib.mila_gpt_parameter_count.restype = C.c_int64
        _lib.mila_gpt_parameter_count.argtypes = [C.c_void_p]
        _lib.mila_gpt_load_parameter.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        _lib.mila_gpt_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _check(rc):
    if rc != 0:
        text = load().mila_host_last_error().decode()
        if rc == capi.MILA_E_INVALID_ARGUMENT:
            raise ValueError(text)
        raise RuntimeError(text)


class Gemma:
    """GemmaTransformer<policy> with synthetic weights on cuda:0."""

    MODES = {"reference": 0, "fused": 1, "graph": 2}

    def __init__(self, policy="bf16", config=None, max_seq=4096, max_prefill=1, seed=1234, device=None, profile=None):
        """device: HIP device ordinal; default = this process's LOCAL_RANK (one replica per GPU under torch.distributed.run).
        profile: synthetic-parameter multipliers {linear_gain, qk_norm_center, post_norm_center, layer_scalar, table_gain}
        (GemmaTransformer::SyntheticProfile; None = the unit-scale generator of SURVEY.md section 8d)"""
        lib = load()
        if device is None:
            from .replicas import local_device
            device = local_device()
        self.device = device
        self.cfg = dict(GEMMA4_12B if config is None else config)
        self.cfg.setdefault("bounded_local_kv", 0)      # 1: SlidingWindowKvCache on the sliding-window layers
        c = GemmaConfigC(**self.cfg)
        self.vocab = self.cfg["vocab_size"]
        self.h = lib.mila_gemma_create(POLICIES[policy], C.byref(c), max_seq, max_prefill, seed, device)
        if not self.h:
            text = lib.mila_host_last_error().decode()
            raise (ValueError if text.startswith("invalid_argument") else RuntimeError)("mila_gemma_create: " + text)
        if profile is not None:
            p = (C.c_float * 5)(*[float(profile[k]) for k in ("linear_gain", "qk_norm_center", "post_norm_center", "layer_scalar", "table_gain")])
            _check(lib.mila_gemma_init_synthetic(self.h, seed, p))

    def close(self):
        if self.h:
            load().mila_gemma_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rewind(self, position):
        """GemmaTransformer::rewindKvCache(position): True when every block accepted (False beyond the fill, or when a bounded ring has evicted what is needed)"""
        lib = load()
        lib.mila_gemma_rewind.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        ok = C.c_int()
        _check(lib.mila_gemma_rewind(self.h, int(position), C.byref(ok)))
        return bool(ok.value)

    def prefill_from(self, tokens, offset):
        """GemmaTransformer::prefillFrom: positions [0, offset) stay resident, tokens[offset:] are prefilled at their positions; logits of the last position"""
        lib = load()
        lib.mila_gemma_prefill_from.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.vocab, dtype=np.float32)
        _check(lib.mila_gemma_prefill_from(self.h, t.ctypes.data, t.size, int(offset), out.ctypes.data))
        return out

    def set_prefill_overlap(self, on):
        """prefill: the chunk's two halves as two kernel sequences on two streams (the second half's attention waits for the first half's K / V rows);
        bf16 and resident-fp8 policies, T % 512 == 0.  Identical bits."""
        lib = load()
        lib.mila_gemma_set_prefill_overlap.argtypes = [C.c_void_p, C.c_int]
        _check(lib.mila_gemma_set_prefill_overlap(self.h, int(bool(on))))

    def save_safetensors(self, path):
        """every parameter in its storage form (bf16, or e4m3 / packed e2m1 + fp32 scales) as a SafeTensors file"""
        _check(load().mila_gemma_save_safetensors(self.h, str(path).encode()))

    def load_safetensors(self, path):
        """load every parameter from a SafeTensors file; bf16 Linear weights are quantized on load under a quantized policy"""
        _check(load().mila_gemma_load_safetensors(self.h, str(path).encode()))

    def save_milabin(self, path):
        """the same tensors in the reference's MILA .bin container (what fromPretrained streams)"""
        _check(load().mila_gemma_save_milabin(self.h, str(path).encode()))

    def load_pretrained(self, path):
        """load every parameter from a MILA .bin or a SafeTensors artifact (sniffed by the leading magic)"""
        _check(load().mila_gemma_load_pretrained(self.h, str(path).encode()))

    def set_resident_prefill_weights(self, on):
        """quantized policies: keep the prefill staging of every layer Linear (fp8 -> bf16, fp4 -> e4m3) resident in HBM (default)
        or re-stage it into scratch on every forward as the reference does; identical bits"""
        _check(load().mila_gemma_set_resident_prefill_weights(self.h, int(bool(on))))

    def memory_stats(self):
        """GemmaTransformer::getRequiredMemory() (from the configuration alone) beside getMemoryStats() (what the built model holds), and the context's scratch high-water
        mark: {"required": {...}, "actual": {...}, "scratch_bytes": n} with device_parameter_bytes / device_state_bytes / host_state_bytes"""
        lib = load()
        lib.mila_gemma_memory_stats.argtypes = [C.c_void_p, C.c_void_p]
        out = (C.c_double * 7)()
        _check(lib.mila_gemma_memory_stats(self.h, out))
        keys = ("device_parameter_bytes", "device_state_bytes", "host_state_bytes")
        return {"required": {k: int(out[i]) for i, k in enumerate(keys)}, "actual": {k: int(out[3 + i]) for i, k in enumerate(keys)}, "scratch_bytes": int(out[6])}

    def graph_node_count(self):
        """nodes of the captured decode graph = launches per token on the graph path (0 before the first graph-mode step)"""
        lib = load()
        lib.mila_gemma_graph_node_count.argtypes = [C.c_void_p, C.c_void_p]
        out = C.c_int64()
        _check(lib.mila_gemma_graph_node_count(self.h, C.byref(out)))
        return out.value

    def resident_staging_bytes(self):
        """bytes of op-owned prefill staging the layer Linears hold right now (fp8 policy: bf16 copies -- 0 while W8A8 is on; fp4 policy: e4m3 copies)"""
        lib = load()
        lib.mila_gemma_resident_staging_bytes.argtypes = [C.c_void_p, C.c_void_p]
        out = C.c_double()
        _check(lib.mila_gemma_resident_staging_bytes(self.h, C.byref(out)))
        return out.value

    def set_fp8_activation_prefill(self, on):
        """fp4 policy: W4A8 prefill on the fp8 matrix cores (default, the reference's default) or the exact-weight bf16 fallback.
        fp8 policy: the OPT-IN W8A8 prefill (the policy's e4m3 weights + per-channel scales on the fp8 matrix cores, per-token e4m3 activations, no bf16 copy of the
        weights; default off -- the reference's arithmetic for PerChannelFp8<> is W8A16)"""
        _check(load().mila_gemma_set_fp8_activation_prefill(self.h, int(bool(on))))

    def set_fused_prefill(self, on):
        """prefill with the fused glue kernels (default, when 1024 < D <= 8192) or one launch per reference op; same bits"""
        _check(load().mila_gemma_set_fused_prefill(self.h, int(bool(on))))

    def prefill(self, tokens, position_offset=0):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.vocab, dtype=np.float32)
        _check(load().mila_gemma_prefill(self.h, t.ctypes.data, t.size, position_offset, out.ctypes.data))
        return out

    def decode(self, token, position, mode="fused"):
        out = np.empty(self.vocab, dtype=np.float32)
        _check(load().mila_gemma_decode(self.h, int(token), int(position), self.MODES[mode], out.ctypes.data))
        return out

    def generate(self, first_token, start_position, n_tokens, mode="graph"):
        """greedy autoregressive generation with the device sampler; returns the sampled token ids"""
        out = np.empty(n_tokens, dtype=np.int32)
        _check(load().mila_gemma_generate(self.h, int(first_token), int(start_position), int(n_tokens), self.MODES[mode], out.ctypes.data))
        return out

    def generate_sampled(self, first_token, start_position, n_tokens, temperature=1.0, top_k=0, top_p=1.0, seed=0, mode="fused"):
        """multinomial generation (softcap + temperature + top-k + top-p on the device, uniform draws from a host mt19937)"""
        out = np.empty(n_tokens, dtype=np.int32)
        _check(load().mila_gemma_generate_sampled(self.h, int(first_token), int(start_position), int(n_tokens), self.MODES[mode], float(temperature),
                                                  int(top_k), float(top_p), int(seed), out.ctypes.data))
        return out

    def time_decode(self, start_position, steps, warmup, mode="graph"):
        out = (C.c_double * 2)()
        _check(load().mila_gemma_time_decode(self.h, start_position, steps, warmup, self.MODES[mode], out))
        return {"wall_ms_per_step": out[0], "device_ms_per_step": out[1]}

    def time_dominant_kernel(self, rounds=3):
        out = (C.c_double * 2)()
        _check(load().mila_gemma_time_dominant_kernel(self.h, rounds, out))
        return {"avg_us": out[0], "bytes": out[1]}

    def time_prefill(self, T, reps=1):
        out = C.c_double()
        _check(load().mila_gemma_time_prefill(self.h, T, reps, C.byref(out)))
        return out.value

    def time_prefill_chunked(self, total_T):
        """device ms of a total_T-token prompt prefilled from an empty cache in chunks of the built prefill size (GemmaTransformer::prefillFrom); the caches
        hold total_T positions afterwards"""
        lib = load()
        lib.mila_gemma_time_prefill_chunked.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        out = C.c_double()
        _check(lib.mila_gemma_time_prefill_chunked(self.h, int(total_T), C.byref(out)))
        return out.value

    def component_names(self):
        """names of the model's components in construction order (each block, its children, temb, rmsn_final, lm_head)"""
        lib = load()
        lib.mila_gemma_component_names.restype = C.c_int64
        lib.mila_gemma_component_names.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        need = lib.mila_gemma_component_names(self.h, None, 0)
        if need < 0:
            raise RuntimeError(lib.mila_host_last_error().decode())
        buf = C.create_string_buffer(need)
        lib.mila_gemma_component_names(self.h, buf, need)
        return buf.value.decode().split("\n")[:-1]

    def info(self, context):
        out = (C.c_double * 4)()
        _check(load().mila_gemma_info(self.h, context, out))
        return {"decode_bytes_per_token": out[0], "weight_bytes": out[1], "linear_params": out[2], "table_params": out[3]}


GENERATE_STATUS = {0: "stop", 1: "length", 2: "context_limit", 3: "cancelled"}      # GenerateStatus / to_string (Core/GenerateStatus.ixx)


class GemmaModel:
    """GemmaModel<Rocm, BF16> (Models/GemmaModel.ixx): fromPretrained / fromSynthetic, then generate()."""

    def __init__(self, handle, device):
        self.h, self.device = handle, device

    @staticmethod
    def _bind():
        lib = load()
        lib.mila_gemma_model_from_pretrained.restype = C.c_void_p
        lib.mila_gemma_model_from_pretrained.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int]
        lib.mila_gemma_model_synthetic.restype = C.c_void_p
        lib.mila_gemma_model_synthetic.argtypes = [C.c_int, C.POINTER(GemmaConfigC), C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_int]
        lib.mila_gemma_model_destroy.argtypes = [C.c_void_p]
        lib.mila_gemma_model_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int64,
                                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        return lib

    @staticmethod
    def _raise(lib, what):
        text = lib.mila_host_last_error().decode()
        raise (ValueError if text.startswith("invalid_argument") else RuntimeError)(what + ": " + text)

    @classmethod
    def from_pretrained(cls, path, policy="bf16", context=4096, prefill_chunk=0, bounded_local_kv=False, device=0):
        """GemmaModel::fromPretrained(path, GemmaModelConfig(context).withWeightQuantization(policy)): the geometry comes from the artifact"""
        lib = cls._bind()
        h = lib.mila_gemma_model_from_pretrained(str(path).encode(), POLICIES[policy], context, prefill_chunk, int(bool(bounded_local_kv)), device)
        if not h:
            cls._raise(lib, "GemmaModel.from_pretrained")
        return cls(h, device)

    @classmethod
    def synthetic(cls, policy="bf16", config=None, context=4096, prefill_chunk=0, seed=1234, profile=None, device=0):
        lib = cls._bind()
        cfg = dict(GEMMA4_12B if config is None else config)
        cfg.setdefault("bounded_local_kv", 0)
        c = GemmaConfigC(**cfg)
        p = None
        if profile is not None:
            p = (C.c_float * 5)(*[float(profile[k]) for k in ("linear_gain", "qk_norm_center", "post_norm_center", "layer_scalar", "table_gain")])
        h = lib.mila_gemma_model_synthetic(POLICIES[policy], C.byref(c), context, prefill_chunk, seed, p, device)
        if not h:
            cls._raise(lib, "GemmaModel.synthetic")
        return cls(h, device)

    def generate(self, prompt, max_new_tokens=None, stop_tokens=(), temperature=1.0, top_k=1, top_p=1.0, seed=None, cancel_after=None):
        """-> (tokens passed to on_token, finish reason as GenerateStatus's to_string, prompt tokens served from the KV caches).
        top_k = 1 is greedy (SamplingParams); seed reseeds the host RNG that draws the sampler's uniform; cancel_after = n raises the client's stop request
        from inside on_token once n tokens were delivered"""
        lib = load()
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        st = np.ascontiguousarray(list(stop_tokens), dtype=np.int32)
        cap = int(max_new_tokens) if max_new_tokens is not None else 1 << 16
        out = np.zeros(max(cap, 1), dtype=np.int32)
        n, status, reused = C.c_int64(), C.c_int32(), C.c_int64()
        _check(lib.mila_gemma_model_generate(self.h, pr.ctypes.data, len(pr), -1 if max_new_tokens is None else int(max_new_tokens), st.ctypes.data if len(st) else None, len(st),
                                             float(temperature), int(top_k), float(top_p), -1 if seed is None else int(seed), out.ctypes.data, len(out), C.byref(n), C.byref(status),
                                             C.byref(reused), -1 if cancel_after is None else int(cancel_after)))
        return out[:min(n.value, len(out))].tolist(), GENERATE_STATUS[status.value], reused.value

    def close(self):
        if self.h:
            load().mila_gemma_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gpt:
    """GptTransformer (GPT-2) on cuda:0; parameters in the order of oracle orc_cpu_gpt2_forward.
    precision "bf16" (BASELINE config 2): parameters and logits are bf16 bit patterns (uint16); "fp32" (BASELINE config 1's model on the device, through the FP32 rows
    of Linear / LayerNorm / MHA / GELU / Residual / LPE): float32 parameters and logits."""

    def __init__(self, vocab, max_seq, C_, L, NH, B, T, precision="bf16"):
        lib = load()
        lib.mila_gpt_create_p.restype = C.c_void_p
        lib.mila_gpt_create_p.argtypes = [C.c_int] + [C.c_int64] * 7
        self.shape = (B, T, vocab)
        self.precision = precision
        self.dtype = {"bf16": np.uint16, "fp32": np.float32}[precision]
        self.h = lib.mila_gpt_create_p({"bf16": 0, "fp32": 1}[precision], vocab, max_seq, C_, L, NH, B, T)
        if not self.h:
            raise (ValueError if b"invalid_argument" in lib.mila_gpt_last_error() else RuntimeError)(lib.mila_gpt_last_error().decode())

    def component_names(self):
        """names of the model's components in construction order (lenc, each block and its children, ln_final, lm_head)"""
        lib = load()
        lib.mila_gpt_component_names.restype = C.c_int64
        lib.mila_gpt_component_names.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        need = lib.mila_gpt_component_names(self.h, None, 0)
        if need < 0:
            raise RuntimeError(lib.mila_gpt_last_error().decode())
        buf = C.create_string_buffer(need)
        lib.mila_gpt_component_names(self.h, buf, need)
        return buf.value.decode().split("\n")[:-1]

    def memory_stats(self):
        """GptTransformer::getRequiredMemory() beside getMemoryStats(): {"required": {...}, "actual": {...}} with device_parameter_bytes / device_state_bytes"""
        lib = load()
        lib.mila_gpt_memory_stats.argtypes = [C.c_void_p, C.c_void_p]
        out = (C.c_double * 4)()
        if lib.mila_gpt_memory_stats(self.h, out):
            raise RuntimeError(lib.mila_gpt_last_error().decode())
        return {"required": {"device_parameter_bytes": int(out[0]), "device_state_bytes": int(out[1])}, "actual": {"device_parameter_bytes": int(out[2]), "device_state_bytes": int(out[3])}}

    def load_parameters(self, params_bf16_bits):
        lib = load()
        assert len(params_bf16_bits) == lib.mila_gpt_parameter_count(self.h)
        for i, p in enumerate(params_bf16_bits):
            p = np.ascontiguousarray(p, dtype=self.dtype)
            rc = lib.mila_gpt_load_parameter(self.h, i, p.ctypes.data, p.nbytes)
            if rc:
                raise ValueError(lib.mila_gpt_last_error().decode())

    def forward(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.shape, dtype=self.dtype)
        ms = C.c_double()
        rc = load().mila_gpt_forward(self.h, t.ctypes.data, out.ctypes.data, C.byref(ms))
        if rc < 0:
            raise RuntimeError(load().mila_gpt_last_error().decode())
        if rc > 0:
            raise IndexError("token index outside vocabulary range (flat position %d)" % (rc - 1))
        self.last_ms = ms.value
        return out

    def forward_timed(self, tokens):
        """forward() without the logits copy to the host (823 MB at config 2): device time in self.last_ms"""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        ms = C.c_double()
        rc = load().mila_gpt_forward(self.h, t.ctypes.data, None, C.byref(ms))
        if rc:
            raise RuntimeError(load().mila_gpt_last_error().decode() if rc < 0 else "token index outside the vocabulary")
        self.last_ms = ms.value
        return ms.value

    def prefill(self, tokens):
        """GptTransformer::prefill: tokens [B, T' <= T] -> logits [B, V] (bf16 bits) of the last position; every block's KV cache is filled"""
        lib = load()
        lib.mila_gpt_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        assert t.ndim == 2 and t.shape[0] == self.shape[0]
        out = np.empty((self.shape[0], self.shape[2]), dtype=self.dtype)
        rc = lib.mila_gpt_prefill(self.h, t.ctypes.data, t.shape[1], out.ctypes.data)
        if rc:
            text = lib.mila_gpt_last_error().decode()
            raise (ValueError if rc == capi.MILA_E_INVALID_ARGUMENT else RuntimeError)(text)
        return out

    def decode(self, tokens, position):
        """GptTransformer::decode: one token per sequence [B] at absolute `position` -> logits [B, V] (bf16 bits)"""
        lib = load()
        lib.mila_gpt_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        t = np.ascontiguousarray(tokens, dtype=np.int32).reshape(-1)
        assert t.size == self.shape[0]
        out = np.empty((self.shape[0], self.shape[2]), dtype=self.dtype)
        rc = lib.mila_gpt_decode(self.h, t.ctypes.data, int(position), out.ctypes.data)
        if rc:
            text = lib.mila_gpt_last_error().decode()
            raise (ValueError if rc == capi.MILA_E_INVALID_ARGUMENT else RuntimeError)(text)
        return out

    def close(self):
        if self.h:
            load().mila_gpt_destroy(self.h)
            self.h = None


def safetensors_list(path):
    """host-only: [(name, dtype, nbytes, shape)], {metadata} of a SafeTensors file as the C++ reader sees it"""
    lib = load()
    need = lib.mila_safetensors_list(str(path).encode(), None, 0)
    if need < 0:
        _check(int(need))
    buf = C.create_string_buffer(int(need) + 1)
    lib.mila_safetensors_list(str(path).encode(), buf, int(need) + 1)
    tensors, meta = [], {}
    for line in buf.value.decode().splitlines():
        if line.startswith("# "):
            k, v = line[2:].split("=", 1)
            meta[k] = v
        else:
            name, dtype, nbytes, shape = (line.split(" ") + [""])[:4]
            tensors.append((name, dtype, int(nbytes), tuple(int(d) for d in shape.split(",") if d)))
    return tensors, meta


def safetensors_copy(src, dst):
    """host-only: rewrite src as dst through the C++ reader and writer"""
    _check(load().mila_safetensors_copy(str(src).encode(), str(dst).encode()))


def _text_call(fn, *args):
    need = fn(*args, None, 0)
    if need < 0:
        _check(int(need))
    buf = C.create_string_buffer(int(need) + 1)
    fn(*args, buf, int(need) + 1)
    return buf.value.decode()


def component_scenarios(device=0):
    """the leaf components (Residual, Swiglu<Gelu>, Rope, GroupedQueryAttention, TokenEmbedding + the tied Linear) used standalone on the GPU,
    checked against the launchers they resolve to, with the reference's lifecycle errors; raises with the failing scenario's name"""
    lib = load()
    lib.mila_component_scenarios.argtypes = [C.c_int]
    _check(lib.mila_component_scenarios(device))


def pretrained_list(path):
    """host-only: ([(name, dtype, nbytes, shape)] in ascending file-offset order, {container, mila_quantization, mila_config})
    of a MILA .bin or SafeTensors file as the C++ PretrainedModelReader sees it"""
    tensors, meta = [], {}
    for line in _text_call(load().mila_pretrained_list, str(path).encode()).splitlines():
        if line.startswith("# "):
            k, v = line[2:].split("=", 1)
            meta[k] = v
        else:
            name, dtype, nbytes, shape = (line.split(" ") + [""])[:4]
            tensors.append((name, dtype, int(nbytes), tuple(int(d) for d in shape.split(",") if d)))
    return tensors, meta


def pretrained_to_milabin(src, dst, metadata_json=None):
    """host-only: rewrite a SafeTensors (or MILA) file as a MILA .bin through the C++ reader and writer"""
    _check(load().mila_pretrained_to_milabin(str(src).encode(), str(dst).encode(), None if metadata_json is None else metadata_json.encode()))


def pretrained_metadata_roundtrip(json_text):
    """host-only: toMetadataJSON(parseMetadataJSON(text))"""
    return _text_call(load().mila_pretrained_metadata_roundtrip, json_text.encode())


class LinearComponent:
    """ONE Linear<Rocm, BF16, policy> of the host mirror (libmila_host: host/src/linear_runner.cpp): load a weight as the reference's loadParameter does (bf16 blob ->
    quantize-on-load under a quantized policy, or the policy's storage form + weight_scale), then forward() at any row count through RocmLinearOp::forward."""

    def __init__(self, policy, K, N, max_rows, bias=False, device=0):
        lib = load()
        lib.mila_linear_create.restype = C.c_void_p
        lib.mila_linear_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
        lib.mila_linear_destroy.argtypes = [C.c_void_p]
        lib.mila_linear_load.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        lib.mila_linear_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mila_linear_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.mila_linear_forward.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        lib.mila_linear_last_error.restype = C.c_char_p
        self.K, self.N, self.policy = K, N, policy
        self.h = lib.mila_linear_create(POLICIES[policy], K, N, max_rows, int(bool(bias)), device)
        if not self.h:
            raise RuntimeError(lib.mila_linear_last_error().decode())

    def _check(self, rc):
        if rc:
            text = load().mila_linear_last_error().decode()
            raise (ValueError if rc == capi.MILA_E_INVALID_ARGUMENT else (TypeError if rc == capi.MILA_E_UNSUPPORTED else RuntimeError))(text)

    def load(self, name, blob):
        b = np.ascontiguousarray(blob)
        self._check(load().mila_linear_load(self.h, name.encode(), b.ctypes.data, b.nbytes))

    def read(self):
        """(stored weight bytes as uint8, scales as float32 or None)"""
        lib = load()
        wb, sb = C.c_int64(), C.c_int64()
        self._check(lib.mila_linear_read(self.h, None, C.byref(wb), None, C.byref(sb)))
        w = np.empty(wb.value, dtype=np.uint8)
        s = np.empty(sb.value // 4, dtype=np.float32)
        self._check(lib.mila_linear_read(self.h, w.ctypes.data, None, s.ctypes.data if sb.value else None, None))
        return w, (s if sb.value else None)

    def set(self, fp8_activation_prefill=None, resident=None):
        self._check(load().mila_linear_set(self.h, -1 if fp8_activation_prefill is None else int(bool(fp8_activation_prefill)), -1 if resident is None else int(bool(resident))))

    def forward(self, x_bf16_bits):
        x = np.ascontiguousarray(x_bf16_bits, dtype=np.uint16)
        M = x.size // self.K
        y = np.empty((M, self.N), dtype=np.uint16)
        self._check(load().mila_linear_forward(self.h, M, x.ctypes.data, y.ctypes.data))
        return y

    def close(self):
        if self.h:
            load().mila_linear_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def linear_install_shared_probe(policy, which):
    """Linear<policy>::installSharedWeight( nullptr ) (which 0) / ( nullptr, nullptr ) (which 1) on an unbuilt component: returns "logic_error", "invalid_argument" or "ok" """
    lib = load()
    lib.mila_linear_install_shared_probe.argtypes = [C.c_int, C.c_int]
    rc = lib.mila_linear_install_shared_probe(POLICIES[policy], which)
    return {0: "ok", capi.MILA_E_UNSUPPORTED: "logic_error", capi.MILA_E_INVALID_ARGUMENT: "invalid_argument"}.get(rc, "error %d" % rc)


class Sampler:
    """RocmSamplingOp (counterpart of CudaSamplingOp<FP32>): sample() = forward + readback; sample_enqueued() = enqueueForward + awaitToken; await_token() alone raises TypeError
    (std::logic_error) when nothing is outstanding"""

    def __init__(self, vocab, softcap=0.0, device=0):
        lib = load()
        lib.mila_sampler_create.restype = C.c_void_p
        lib.mila_sampler_create.argtypes = [C.c_int64, C.c_float, C.c_int]
        lib.mila_sampler_destroy.argtypes = [C.c_void_p]
        lib.mila_sampler_set_logits.argtypes = [C.c_void_p, C.c_void_p]
        lib.mila_sampler_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p]
        lib.mila_linear_last_error.restype = C.c_char_p
        self.vocab = vocab
        self.h = lib.mila_sampler_create(vocab, softcap, device)
        if not self.h:
            raise RuntimeError(lib.mila_linear_last_error().decode())

    def set_logits(self, logits):
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        assert lg.size == self.vocab
        self._keep = lg          # the copy is asynchronous on the context stream
        if load().mila_sampler_set_logits(self.h, lg.ctypes.data):
            raise RuntimeError(load().mila_linear_last_error().decode())

    def _run(self, mode, temperature, top_k, top_p, r):
        tok = C.c_int32()
        rc = load().mila_sampler_sample(self.h, mode, temperature, top_k, top_p, r, C.byref(tok))
        if rc:
            text = load().mila_linear_last_error().decode()
            raise (TypeError if rc == capi.MILA_E_UNSUPPORTED else (ValueError if rc == capi.MILA_E_INVALID_ARGUMENT else RuntimeError))(text)
        return tok.value

    def sample(self, temperature=1.0, top_k=0, top_p=1.0, r=0.5):
        return self._run(0, temperature, top_k, top_p, r)

    def sample_enqueued(self, temperature=1.0, top_k=0, top_p=1.0, r=0.5):
        return self._run(1, temperature, top_k, top_p, r)

    def await_token(self):
        return self._run(2, 0.0, 0, 1.0, 0.0)

    def close(self):
        if self.h:
            load().mila_sampler_destroy(self.h)
            self.h = None


class activation_tap:
    """context manager: records the per-token e4m3 activations (bytes + scales) every Linear on an fp8 x fp8 prefill path consumed while it is open
    (Compute::ActivationTap; test instrument) -> .records = [(M, K, N, x8 [M, K] uint8, ts [M] float32)] in call order"""

    def __enter__(self):
        lib = load()
        lib.mila_linear_tap_count.restype = C.c_int64
        lib.mila_linear_tap_get.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mila_linear_tap_begin()
        self.records = []
        return self

    def __exit__(self, *exc):
        lib = load()
        lib.mila_linear_tap_end()
        for i in range(lib.mila_linear_tap_count()):
            dims = (C.c_int32 * 3)()
            lib.mila_linear_tap_get(i, dims, None, None)
            M, K, N = dims[0], dims[1], dims[2]
            x8, ts = np.empty((M, K), dtype=np.uint8), np.empty(M, dtype=np.float32)
            lib.mila_linear_tap_get(i, None, x8.ctypes.data, ts.ctypes.data)
            self.records.append((M, K, N, x8, ts))
        return False

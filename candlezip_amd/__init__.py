"""candlezip_amd — host-side mirror (ctypes) of the MI355X-native candlezip hot path.

The product is libcandlezip_hip.so (candlezip_amd/csrc, C ABI in include/candlezip.h).  This module only marshals
arguments; it mirrors the reference's session trait (src/models.rs:28-33) and token loops so that tests read like the
reference's call sites.  There is no CPU fallback: without the built library, or without a gfx950 device, every
compute call raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CANDLEZIP_AMD_LIB") or os.path.join(_HERE, "lib", "libcandlezip_hip.so")   # override: tuning builds only
AC_CDF_TOTAL = 1 << 30
FLAG_PRODUCER_HIP = 1 << 15     # include/candlezip.h CZ_FLAG_PRODUCER_HIP

BACKEND_SMOLLM, BACKEND_RWKV7 = 0, 1
_STATUS = {0: "CZ_OK", 1: "CZ_ERR_ARG", 2: "CZ_ERR_IO", 3: "CZ_ERR_HIP", 4: "CZ_ERR_STATE", 5: "CZ_ERR_UNSUPPORTED", 6: "CZ_ERR_FORMAT"}


class CandlezipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{_STATUS.get(status, status)}: {msg}")
        self.status = status


class LlamaConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "vocab d_model n_layers n_heads n_kv_heads d_ffn max_pos".split()] + \
               [("rope_theta", C.c_float), ("rms_eps", C.c_float)]


class Rwkv7Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "vocab hidden n_layers head_dim d_ffn a_rank decay_rank v_rank gate_rank".split()] + \
               [("norm_eps", C.c_float)]


class ModelDesc(C.Structure):
    _fields_ = [("backend", C.c_int), ("weight_paths", C.POINTER(C.c_char_p)), ("n_weight_paths", C.c_size_t),
                ("config_path", C.c_char_p), ("synthetic_seed", C.c_uint64), ("synthetic_embed_std", C.c_float),
                ("llama", LlamaConfig), ("rwkv", Rwkv7Config), ("max_streams", C.c_uint32), ("round_f32_weights_to_bf16", C.c_uint32)]


class Header(C.Structure):
    _fields_ = [("bos_token_id", C.c_uint32), ("token_count", C.c_uint64), ("orig_len_bytes", C.c_uint64),
                ("model_hash16", C.c_uint8 * 16), ("tokenizer_hash16", C.c_uint8 * 16), ("orig_hash16", C.c_uint8 * 16),
                ("reserved_flags", C.c_uint32), ("context_window", C.c_uint32), ("vocab_size", C.c_uint32),
                ("model_file_repr_len", C.c_uint32), ("reprime_interval", C.c_uint32)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


def smollm2_135m_config():
    """HuggingFaceTB/SmolLM2-135M config.json (vocab confirmed by the reference's .canz headers)"""
    return LlamaConfig(49152, 576, 30, 9, 3, 1536, 8192, 100000.0, 1e-5)


def tiny_llama_config(vocab=512, d_model=128, n_layers=2, n_heads=2, n_kv_heads=1, d_ffn=256, max_pos=2048):
    return LlamaConfig(vocab, d_model, n_layers, n_heads, n_kv_heads, d_ffn, max_pos, 100000.0, 1e-5)


def rwkv7_0b1_config():
    return Rwkv7Config(65536, 768, 12, 64, 3072, 64, 64, 32, 128, 1e-5)


def tiny_rwkv7_config(vocab=512, hidden=128, n_layers=2, d_ffn=512):
    return Rwkv7Config(vocab, hidden, n_layers, 64, d_ffn, 64, 64, 32, 128, 1e-5)


EXPORTS = [  # every symbol include/candlezip.h declares
    "cz_session_open", "cz_session_close", "cz_session_fork", "cz_vocab_size", "cz_max_context_length", "cz_index_pos", "cz_step_logits",
    "cz_reprime_logits", "cz_encode_tokens", "cz_encode_tokens_batch", "cz_encode_tokens_range_batch", "cz_decode_tokens", "cz_decode_tokens_batch",
    "cz_decode_begin_batch", "cz_decode_begin_partial_batch", "cz_decode_feed_batch", "cz_decode_advance", "cz_decode_fetch_batch", "cz_decode_end",
    "cz_ac_encoder_new", "cz_ac_encoder_push", "cz_ac_encoder_finish", "cz_ac_encoder_free",
    "cz_ac_encode_pairs", "cz_free", "cz_write_header_v2", "cz_read_header_v2", "cz_flags_pack", "cz_skip_gate_section",
    "cz_blake3_bytes_bin16", "cz_blake3_file_bin16", "cz_blake3_files_bin16", "cz_rwkv_tokenizer_open",
    "cz_rwkv_tokenizer_close", "cz_rwkv_plan_symbols", "cz_rwkv_detok", "cz_bpe_tokenizer_open", "cz_bpe_tokenizer_close",
    "cz_bpe_vocab_size", "cz_bpe_encode", "cz_bpe_decode", "cz_xe_bits_over_span", "cz_xe_bits_batch", "cz_last_error", "cz_profile_enable",
    "cz_profile_read", "cz_synchronize",
]

_lib = None


def lib():
    """load libcandlezip_hip.so — raises loudly when it has not been built (no fallback of any kind)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). candlezip_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64
    sig = {
        "cz_session_open": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(vp)]),
        "cz_session_close": (None, [vp]), "cz_session_fork": (C.c_int, [vp, C.POINTER(vp)]), "cz_vocab_size": (sz, [vp]), "cz_max_context_length": (sz, [vp]), "cz_index_pos": (sz, [vp]),
        "cz_step_logits": (C.c_int, [vp, u32, vp]), "cz_reprime_logits": (C.c_int, [vp, vp, sz, vp]),
        "cz_encode_tokens": (C.c_int, [vp, vp, sz, u32, u32, vp, vp]),
        "cz_encode_tokens_batch": (C.c_int, [vp, sz, vp, vp, u32, u32, vp, vp]),
        "cz_encode_tokens_range_batch": (C.c_int, [vp, sz, vp, vp, u32, u32, vp, vp, vp, vp]),
        "cz_decode_begin_batch": (C.c_int, [vp, sz, vp, vp, vp, u32, u32, u32]),
        "cz_decode_begin_partial_batch": (C.c_int, [vp, sz, vp, vp, vp, vp, u32, u32, u32]),
        "cz_decode_feed_batch": (C.c_int, [vp, sz, vp, vp, C.c_int]),
        "cz_ac_encoder_new": (C.c_int, [C.POINTER(vp)]), "cz_ac_encoder_free": (None, [vp]),
        "cz_ac_encoder_push": (C.c_int, [vp, vp, vp, sz, C.POINTER(vp), C.POINTER(sz)]),
        "cz_ac_encoder_finish": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(u64)]),
        "cz_decode_advance": (C.c_int, [vp, sz, C.POINTER(sz), C.POINTER(sz)]),
        "cz_decode_fetch_batch": (C.c_int, [vp, sz, vp]), "cz_decode_end": (C.c_int, [vp]),
        "cz_decode_tokens": (C.c_int, [vp, vp, sz, sz, u32, u32, u32, vp]),
        "cz_decode_tokens_batch": (C.c_int, [vp, sz, vp, vp, vp, u32, u32, u32, vp]),
        "cz_ac_encode_pairs": (C.c_int, [vp, vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(u64)]),
        "cz_free": (None, [vp]),
        "cz_write_header_v2": (sz, [C.POINTER(Header), vp, vp, sz]),
        "cz_read_header_v2": (C.c_int, [vp, sz, C.POINTER(Header), C.POINTER(sz), C.POINTER(sz)]),
        "cz_flags_pack": (u32, [C.c_int, C.c_int, C.c_int, u32]),
        "cz_skip_gate_section": (C.c_int, [vp, sz, C.POINTER(sz)]),
        "cz_blake3_bytes_bin16": (None, [vp, sz, vp]), "cz_blake3_file_bin16": (C.c_int, [C.c_char_p, vp]),
        "cz_blake3_files_bin16": (C.c_int, [C.POINTER(C.c_char_p), sz, vp]),
        "cz_last_error": (C.c_char_p, []),
        "cz_xe_bits_over_span": (C.c_int, [vp, vp, sz, vp, sz, vp, sz, C.POINTER(C.c_double)]),
        "cz_xe_bits_batch": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, vp]),
        "cz_bpe_tokenizer_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]), "cz_bpe_tokenizer_close": (None, [vp]), "cz_bpe_vocab_size": (sz, [vp]),
        "cz_bpe_encode": (C.c_int, [vp, vp, sz, C.POINTER(vp), C.POINTER(sz)]),
        "cz_bpe_decode": (C.c_int, [vp, vp, sz, C.c_int, C.POINTER(vp), C.POINTER(sz)]),
        "cz_rwkv_tokenizer_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]), "cz_rwkv_tokenizer_close": (None, [vp]),
        "cz_rwkv_plan_symbols": (C.c_int, [vp, vp, sz, u32, vp, sz, C.POINTER(sz)]),
        "cz_rwkv_detok": (C.c_int, [vp, vp, sz, u32, C.POINTER(vp), C.POINTER(sz)]),
        "cz_profile_enable": (C.c_int, [vp, C.c_int]), "cz_profile_read": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
        "cz_synchronize": (C.c_int, [vp]),
        "cz_debug_set": (C.c_int, [vp, C.c_char_p, C.c_long]),
        "cz_debug_scalar": (C.c_int, [C.c_int, vp, vp, vp, C.c_int]), "cz_debug_f64": (C.c_int, [C.c_int, vp, vp, vp, C.c_int]),
        "cz_debug_gemm": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "cz_debug_rmsnorm": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_float]),
        "cz_debug_cdf_pairs": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "cz_debug_decode_symbols": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, sz, vp, C.c_int]),
        "cz_debug_get": (C.c_long, [vp, C.c_char_p]),
        "cz_debug_attention": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "cz_debug_attn_variant": (None, [C.c_int]),
        "cz_debug_expf_exhaustive": (C.c_int, [vp]),
    }
    for name, (res, args) in sig.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
    _lib = L
    return L


def reset_process_knobs():
    """every process-wide tuning knob back to the value the library ships with (cz_debug_set "reset_process_knobs"; needs no session)"""
    _check(lib().cz_debug_set(None, b"reset_process_knobs", 0))


def _check(rc):
    if rc != 0:
        raise CandlezipError(rc, (lib().cz_last_error() or b"").decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data if a is not None else None


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


class Session:
    """mirror of `trait LanguageModelSession` (src/models.rs:28-33) + the fused encode/decode loops"""

    def __init__(self, backend="smollm", config=None, weight_paths=(), config_path=None, seed=0x5EED, embed_std=0.02,
                 device=0, max_streams=64, round_f32_weights_to_bf16=False):
        L = lib()
        d = ModelDesc()
        d.backend = {"smollm": BACKEND_SMOLLM, "rwkv7": BACKEND_RWKV7}[backend.lower()]     # --backend (src/main.rs:1799)
        self._paths = [p.encode() for p in weight_paths]
        if self._paths:
            arr = (C.c_char_p * len(self._paths))(*self._paths)
            d.weight_paths = C.cast(arr, C.POINTER(C.c_char_p))
            self._arr = arr
        d.n_weight_paths = len(self._paths)
        d.config_path = config_path.encode() if config_path else None
        d.synthetic_seed, d.synthetic_embed_std, d.max_streams = seed, embed_std, max_streams
        d.round_f32_weights_to_bf16 = 1 if round_f32_weights_to_bf16 else 0
        if d.backend == BACKEND_SMOLLM:
            d.llama = config if config is not None else smollm2_135m_config()
        else:
            d.rwkv = config if config is not None else rwkv7_0b1_config()
        self.backend = backend.lower()
        self.h = C.c_void_p()
        _check(L.cz_session_open(C.byref(d), device, C.byref(self.h)))
        self.V = int(L.cz_vocab_size(self.h))

    def fork(self):
        """a second, independent session on the same device weights (cz_session_fork); close it before this one"""
        f = object.__new__(Session)
        f.h = C.c_void_p()
        _check(lib().cz_session_fork(self.h, C.byref(f.h)))
        f.V = self.V
        f._parent = self                                         # keeps the weights' owner alive
        return f

    def close(self):
        if getattr(self, "h", None):
            lib().cz_session_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- trait methods
    def vocab_size(self):
        return self.V

    def max_context_length(self):
        return int(lib().cz_max_context_length(self.h))

    def index_pos(self):
        return int(lib().cz_index_pos(self.h))

    def step_logits_tensor(self, token_id):
        out = np.empty(self.V, np.float32)
        _check(lib().cz_step_logits(self.h, int(token_id), _p(out)))
        return out

    def reprime_with_history_and_get_last_logits_tensor(self, history):
        history = np.ascontiguousarray(history, np.uint32)
        out = np.empty(self.V, np.float32)
        _check(lib().cz_reprime_logits(self.h, _p(history), history.size, _p(out)))
        return out

    # ---- fused loops
    def encode_tokens_batch(self, ids_list, context=512, reprime_interval=512):
        ids_list = [np.ascontiguousarray(a, np.uint32) for a in ids_list]
        n = np.array([a.size for a in ids_list], np.uint64)
        los = [np.zeros(max(a.size - 1, 1), np.uint32) for a in ids_list]
        his = [np.zeros(max(a.size - 1, 1), np.uint32) for a in ids_list]
        _check(lib().cz_encode_tokens_batch(self.h, len(ids_list), _ptr_array(ids_list), _p(n), context, reprime_interval,
                                            _ptr_array(los), _ptr_array(his)))
        return [(lo[:a.size - 1], hi[:a.size - 1]) for a, lo, hi in zip(ids_list, los, his)]

    def encode_tokens(self, ids, context=512, reprime_interval=512):
        return self.encode_tokens_batch([ids], context, reprime_interval)[0]

    def encode_tokens_range_batch(self, ids_list, ranges, context=512, reprime_interval=512):
        """pairs of the coded tokens i in [b, e) of every stream (ranges = [(b, e), ...]); index 0 of the result is token b"""
        ids_list = [np.ascontiguousarray(a, np.uint32) for a in ids_list]
        n = np.array([a.size for a in ids_list], np.uint64)
        ib = np.array([r[0] for r in ranges], np.uint64); ie = np.array([r[1] for r in ranges], np.uint64)
        los = [np.zeros(max(int(e) - int(b), 1), np.uint32) for b, e in ranges]
        his = [np.zeros(max(int(e) - int(b), 1), np.uint32) for b, e in ranges]
        _check(lib().cz_encode_tokens_range_batch(self.h, len(ids_list), _ptr_array(ids_list), _p(n), context, reprime_interval, _p(ib), _p(ie),
                                                  _ptr_array(los), _ptr_array(his)))
        return [(lo[:int(e) - int(b)], hi[:int(e) - int(b)]) for (b, e), lo, hi in zip(ranges, los, his)]

    # ---- the decode loop in slices (cz_decode_begin_batch / advance / fetch / end)
    def decode_begin_batch(self, payloads, token_counts, context_window=512, reprime_interval=512, bos=0):
        bufs = [np.frombuffer(bytes(p), np.uint8).copy() if len(p) else np.zeros(1, np.uint8) for p in payloads]
        lens = np.array([len(p) for p in payloads], np.uint64)
        tcs = np.array(token_counts, np.uint64)
        self._run_tcs = [int(t) for t in token_counts]
        _check(lib().cz_decode_begin_batch(self.h, len(payloads), _ptr_array(bufs), _p(lens), _p(tcs), context_window, reprime_interval, bos))

    def decode_begin_partial_batch(self, payloads, payload_caps, token_counts, context_window=512, reprime_interval=512, bos=0):
        """begin with the bytes at hand; more arrive through decode_feed_batch (payload_caps = room per stream)"""
        bufs = [np.frombuffer(bytes(p), np.uint8).copy() if len(p) else np.zeros(1, np.uint8) for p in payloads]
        lens = np.array([len(p) for p in payloads], np.uint64)
        caps = np.array(payload_caps, np.uint64); tcs = np.array(token_counts, np.uint64)
        self._run_tcs = [int(t) for t in token_counts]
        _check(lib().cz_decode_begin_partial_batch(self.h, len(payloads), _ptr_array(bufs), _p(lens), _p(caps), _p(tcs), context_window, reprime_interval, bos))

    def decode_feed_batch(self, chunks, final=False):
        bufs = [np.frombuffer(bytes(c), np.uint8).copy() if len(c) else np.zeros(1, np.uint8) for c in chunks]
        lens = np.array([len(c) for c in chunks], np.uint64)
        _check(lib().cz_decode_feed_batch(self.h, len(chunks), _ptr_array(bufs), _p(lens), 1 if final else 0))

    def decode_advance(self, n_tokens):
        """decode up to n_tokens more tokens per stream; returns (position, remaining) of the longest stream"""
        pos, rem = C.c_size_t(), C.c_size_t()
        _check(lib().cz_decode_advance(self.h, int(n_tokens), C.byref(pos), C.byref(rem)))
        return pos.value, rem.value

    def decode_fetch_batch(self):
        outs = [np.zeros(t + 1, np.uint32) for t in self._run_tcs]
        _check(lib().cz_decode_fetch_batch(self.h, len(outs), _ptr_array(outs)))
        return outs

    def decode_end(self):
        _check(lib().cz_decode_end(self.h))

    def decode_tokens_batch(self, payloads, token_counts, context_window=512, reprime_interval=512, bos=0):
        bufs = [np.frombuffer(bytes(p), np.uint8).copy() if len(p) else np.zeros(1, np.uint8) for p in payloads]
        lens = np.array([len(p) for p in payloads], np.uint64)
        tcs = np.array(token_counts, np.uint64)
        outs = [np.zeros(int(t) + 1, np.uint32) for t in token_counts]
        _check(lib().cz_decode_tokens_batch(self.h, len(payloads), _ptr_array(bufs), _p(lens), _p(tcs), context_window,
                                            reprime_interval, bos, _ptr_array(outs)))
        return outs

    def decode_tokens(self, payload, token_count, context_window=512, reprime_interval=512, bos=0):
        return self.decode_tokens_batch([payload], [token_count], context_window, reprime_interval, bos)[0]

    # ---- cross_entropy_bits_over_span (src/main.rs:1725-1751)
    def xe_bits_over_span(self, history, targets, hint=()):
        return float(self.xe_bits_batch([history], [targets], [hint])[0])

    def xe_bits_batch(self, histories, targets, hints=None):
        n = len(targets)
        hints = hints if hints is not None else [()] * n
        hs = [np.ascontiguousarray(h, np.uint32) for h in histories]; gs = [np.ascontiguousarray(g, np.uint32) for g in hints]
        ts = [np.ascontiguousarray(t, np.uint32) for t in targets]
        pad = lambda arrs: [a if a.size else np.zeros(1, np.uint32) for a in arrs]
        out = np.zeros(max(n, 1), np.float64)
        lens = lambda arrs: np.array([a.size for a in arrs] or [0], np.uint64)
        hp, gp, tp = pad(hs), pad(gs), pad(ts)
        hl, gl, tl = lens(hs), lens(gs), lens(ts)                  # (kept alive across the call)
        _check(lib().cz_xe_bits_batch(self.h, n, _ptr_array(hp), _p(hl), _ptr_array(gp), _p(gl), _ptr_array(tp), _p(tl), _p(out)))
        return out[:n]

    # ---- measurement
    KINDS = ["gemm_bf16w_f32_mfma", "gemm_skinny_f32_mfma", "attention_f32_mfma", "rmsnorm", "rope", "swiglu", "embed", "cdf_pairs",
             "decode_symbol", "rwkv7_step", "dec_gemm_plain", "dec_attention", "dec_rmsnorm", "dec_gemm_qkv_rope", "dec_gemm_swiglu",
             "dec_gemm_lm_head", "gemm_bf16w_f32_mfma_n64", "rwkv7_state", "rwkv7_prep", "rwkv7_gemm", "rwkv7_scan"]

    def profile_enable(self, on=True, only=None):
        """HIP-event timing per kernel family on the library's stream; `only` = list of family names to time"""
        v = 0
        if on:
            v = 1
            if only:
                mask = 0
                for k in only:
                    mask |= 1 << self.KINDS.index(k)
                v = 1 | (mask << 1)
        _check(lib().cz_profile_enable(self.h, v))

    def profile_read(self):
        arr = (KernelStat * 32)()
        n = C.c_size_t()
        _check(lib().cz_profile_read(self.h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): dict(launches=int(arr[i].launches), total_ms=arr[i].total_ms, flops=arr[i].flops, bytes=arr[i].bytes)
                for i in range(n.value)}

    def synchronize(self):
        _check(lib().cz_synchronize(self.h))

    def debug_set(self, key, value):
        _check(lib().cz_debug_set(self.h, key.encode(), int(value)))

    def debug_get(self, key):
        return int(lib().cz_debug_get(self.h, key.encode()))


# ---- host-side coder / container (reference: ArithmeticEncoder, write/read_header_v2, blake3 helpers)
def ac_encode_pairs(c_lo, c_hi):
    c_lo = np.ascontiguousarray(c_lo, np.uint32)
    c_hi = np.ascontiguousarray(c_hi, np.uint32)
    pay, ln, zw = C.c_void_p(), C.c_size_t(), C.c_uint64()
    _check(lib().cz_ac_encode_pairs(_p(c_lo), _p(c_hi), c_lo.size, C.byref(pay), C.byref(ln), C.byref(zw)))
    data = C.string_at(pay, ln.value) if ln.value else b""
    lib().cz_free(pay)
    return data, int(zw.value)


class AcEncoder:
    """ArithmeticEncoder fed in pieces (cz_ac_encoder_*): push(c_lo, c_hi) -> the bytes that became final, finish() -> (last bytes, zero_width)"""

    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().cz_ac_encoder_new(C.byref(self.h)))

    def push(self, c_lo, c_hi) -> bytes:
        c_lo = np.ascontiguousarray(c_lo, np.uint32); c_hi = np.ascontiguousarray(c_hi, np.uint32)
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().cz_ac_encoder_push(self.h, _p(c_lo), _p(c_hi), c_lo.size, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def finish(self):
        p, n, tot, zw = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_uint64()
        _check(lib().cz_ac_encoder_finish(self.h, C.byref(p), C.byref(n), C.byref(tot), C.byref(zw)))
        return (C.string_at(p, n.value) if n.value else b""), int(zw.value)

    def close(self):
        if getattr(self, "h", None):
            lib().cz_ac_encoder_free(self.h)
            self.h = None

    __del__ = close


def write_header(h: Header, model_file_repr: bytes) -> bytes:
    h.model_file_repr_len = len(model_file_repr)
    out = np.zeros(128 + len(model_file_repr), np.uint8)
    r = np.frombuffer(model_file_repr, np.uint8) if model_file_repr else np.zeros(1, np.uint8)
    n = lib().cz_write_header_v2(C.byref(h), _p(r), _p(out), out.size)
    if n == 0:
        raise CandlezipError(1, "header buffer too small")
    return out[:n].tobytes()


def read_header(buf: bytes):
    a = np.frombuffer(buf, np.uint8)
    h, off, used = Header(), C.c_size_t(), C.c_size_t()
    _check(lib().cz_read_header_v2(_p(a), len(buf), C.byref(h), C.byref(off), C.byref(used)))
    return h, bytes(buf[off.value:off.value + h.model_file_repr_len]), used.value


class BpeTokenizer:
    """HF byte-level BPE tokenizer.json, in-process (the `tokenizers` crate calls of src/main.rs:1812, 1852, 2654)"""

    def __init__(self, tokenizer_json_path):
        h = C.c_void_p()
        _check(lib().cz_bpe_tokenizer_open(os.fsencode(tokenizer_json_path), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().cz_bpe_tokenizer_close(self.h)
            self.h = None

    __del__ = close

    def vocab_size(self):
        return int(lib().cz_bpe_vocab_size(self.h))

    def encode(self, data: bytes):
        """ids of String::from_utf8_lossy(data), no special tokens added (tok.encode(text, false))"""
        a = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().cz_bpe_encode(self.h, a.ctypes.data, len(data), C.byref(out), C.byref(n)))
        ids = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), (max(n.value, 1),))[: n.value].copy()
        lib().cz_free(out)
        return ids

    def decode(self, ids, skip_special_tokens=True) -> bytes:
        a = np.ascontiguousarray(ids, np.uint32)
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().cz_bpe_decode(self.h, a.ctypes.data if a.size else None, a.size, 1 if skip_special_tokens else 0, C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value) if n.value else b""
        lib().cz_free(out)
        return data


class WorldTokenizer:
    """RWKV "World" tokenizer (candle_rwkv7/src/models/rwkv7.rs:546-610) + plan_rwkv_symbols / rwkv_detok_with_literals
    (src/main.rs:833-895), in-process through the C ABI (cz_rwkv_*)"""

    def __init__(self, vocab_json_path):
        h = C.c_void_p()
        _check(lib().cz_rwkv_tokenizer_open(os.fsencode(vocab_json_path), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().cz_rwkv_tokenizer_close(self.h)
            self.h = None

    __del__ = close

    def plan_symbols(self, data: bytes, vocab: int, ids_cap=None):
        """symbols of `data`: token ids, or vocab + byte literals (uncovered tail / complete fallback)"""
        buf = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
        cap = len(data) if ids_cap is None else ids_cap
        out = np.zeros(max(cap, 1), np.uint32); n = C.c_size_t()
        _check(lib().cz_rwkv_plan_symbols(self.h, buf.ctypes.data, len(data), vocab, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def detok(self, ids, vocab: int) -> bytes:
        a = np.ascontiguousarray(ids, np.uint32); out = C.c_void_p(); n = C.c_size_t()
        _check(lib().cz_rwkv_detok(self.h, a.ctypes.data if a.size else None, a.size, vocab, C.byref(out), C.byref(n)))
        data = C.string_at(out, n.value) if n.value else b""
        lib().cz_free(out)
        return data


def blake3_16(data: bytes) -> bytes:
    a = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(16, np.uint8)
    lib().cz_blake3_bytes_bin16(_p(a), len(data), _p(out))
    return out.tobytes()

/*
 * candlezip.h — C ABI of the MI355X-native candlezip hot path (libcandlezip_hip.so).
 *
 * Drop-in boundary for the reference's model-session / coder interface (turtle261/candlezip @ 2025-10-10).
 * Each entry point cites the reference interface it replaces (paths relative to the reference root).
 * Conventions: opaque handles, plain pointers and sizes, int status codes, no exceptions across the ABI,
 * caller allocates every output, the library owns all device memory.  Thread-compatible: one handle is used
 * by one thread at a time (the reference session is an owned `&mut` object, src/models.rs:28-33).
 * The reference-side binding (Rust `impl LanguageModelSession`) is shown in INTEGRATION.md.
 *
 * There is NO CPU fallback: every compute entry point returns CZ_ERR_HIP when no gfx950 device is usable.
 */
#ifndef CANDLEZIP_H
#define CANDLEZIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum cz_status {
    CZ_OK = 0,
    CZ_ERR_ARG = 1,          /* bad argument (null, size, token >= vocab, empty history ...) */
    CZ_ERR_IO = 2,           /* file missing / malformed safetensors / config.json */
    CZ_ERR_HIP = 3,          /* HIP runtime error or no gfx950 device */
    CZ_ERR_STATE = 4,        /* position would exceed max_position_embeddings, stream exhausted ... */
    CZ_ERR_UNSUPPORTED = 5,  /* model shape outside what the kernels support (head_dim != 64 ...) */
    CZ_ERR_FORMAT = 6        /* container: bad magic / version / truncated */
} cz_status;

typedef enum cz_backend { CZ_BACKEND_SMOLLM = 0, CZ_BACKEND_RWKV7 = 1 } cz_backend;   /* --backend, src/main.rs:38-39 */

/* LLaMA-family hyper-parameters (HF config.json keys; candle LlamaConfig, src/models.rs:50-53). head_dim = 64. */
typedef struct cz_llama_config {
    int32_t vocab, d_model, n_layers, n_heads, n_kv_heads, d_ffn, max_pos;
    float rope_theta, rms_eps;
} cz_llama_config;

/* RWKV7 hyper-parameters (candle_rwkv7/src/models/rwkv7.rs:7-32). head_dim = 64. */
typedef struct cz_rwkv7_config {
    int32_t vocab, hidden, n_layers, head_dim, d_ffn, a_rank, decay_rank, v_rank, gate_rank;
    float norm_eps;
} cz_rwkv7_config;

typedef struct cz_model_desc {
    cz_backend backend;
    /* weights: safetensors files (src/models.rs:48 `weights_paths`, :132 `model_path`) ... */
    const char* const* weight_paths;
    size_t n_weight_paths;
    const char* config_path;            /* config.json (src/models.rs:49-52, :133-136); NULL => use `llama` / `rwkv` below */
    /* ... or, when n_weight_paths == 0, seeded synthetic weights (no checkpoints exist offline) */
    uint64_t synthetic_seed;
    float synthetic_embed_std;
    cz_llama_config llama;
    cz_rwkv7_config rwkv;
    uint32_t max_streams;               /* capacity of one lock-step decode group (0 => 64, at most 1024); KV cache: 94 MB per stream at the SmolLM2-135M shape */
    /* Weights live in HBM as bf16.  An F32 tensor whose values bf16 cannot hold (the converted RWKV7 checkpoints) is refused
     * (CZ_ERR_UNSUPPORTED) unless this is non-zero: then it is rounded to nearest-even bf16 — a slightly different model than the
     * reference's f32 one (other payloads, still self-consistent between compress and decompress).  Explicit opt-in only. */
    uint32_t round_f32_weights_to_bf16;
} cz_model_desc;

typedef struct cz_session cz_session;

/* SmolLmSession::load (src/models.rs:48-61) / Rwkv7Session::load (src/models.rs:132-146) */
cz_status cz_session_open(const cz_model_desc* desc, int device_ordinal, cz_session** out);
void cz_session_close(cz_session* s);
/* A second, independent session on the SAME device weights (own stream, KV cache / recurrent state, buffers).  The reference gets a
 * second session by loading the model again (agent path, src/main.rs:1943-1956); a fork costs no weight memory and lets another
 * host thread compress while this one decompresses.  Close every fork before its parent: a fork that outlives it fails every call
 * (CZ_ERR_STATE) and can only be closed. */
cz_status cz_session_fork(cz_session* parent, cz_session** out);

/* trait LanguageModelSession (src/models.rs:28-33), 1:1 */
size_t cz_vocab_size(const cz_session* s);                      /* vocab_size()            :29 */
size_t cz_max_context_length(const cz_session* s);              /* max_context_length()    :30 (512 / usize::MAX) */
size_t cz_index_pos(const cz_session* s);                       /* SmolLmSession::index_pos :65 */
cz_status cz_step_logits(cz_session* s, uint32_t token, float* out_logits /* [vocab] */);          /* :31, :92-103, :150-160 */
cz_status cz_reprime_logits(cz_session* s, const uint32_t* history, size_t n, float* out_logits);  /* :32, :104-119, :161-180 */

/* Fused fast paths: the token loops of encode_file / decode_file with the agent off, logits never leave HBM.
 *
 * cz_encode_tokens  replaces src/main.rs:1913-1916, 1935, 1979, 2275-2325, 2344-2349 : ids[0] is BOS, n counts BOS.
 *   Writes for every coded token i in [0, n-1) the interval (c_lo, c_hi) the reference passes to
 *   ArithmeticEncoder::encode_counts (total 2^30).  Teacher-forced and segment-parallel on the GPU; the serial
 *   coder (cz_ac_encode_pairs) then turns the pairs into the payload.  `context` / `reprime_interval` are the CLI
 *   values (src/main.rs:104-109); both are ignored for RWKV7 (no re-prime, src/main.rs:2275).
 * cz_decode_tokens  replaces src/main.rs:2485, 2506, 2528-2541, 2621-2628 (SmolLM) / 2812-2834 (RWKV7):
 *   out_ids receives token_count+1 ids (BOS first).  The arithmetic-decoder state lives on the device.
 * The *_batch forms run n_streams independent files/chunks: encode packs all their segments into shared GEMMs,
 * decode steps the streams in lock-step (this is where decode throughput comes from). */
cz_status cz_encode_tokens(cz_session* s, const uint32_t* ids, size_t n, uint32_t context, uint32_t reprime_interval,
                           uint32_t* out_c_lo, uint32_t* out_c_hi);
cz_status cz_encode_tokens_batch(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n,
                                 uint32_t context, uint32_t reprime_interval,
                                 uint32_t* const* out_c_lo, uint32_t* const* out_c_hi);
cz_status cz_decode_tokens(cz_session* s, const uint8_t* payload, size_t len, size_t token_count,
                           uint32_t context_window, uint32_t reprime_interval, uint32_t bos, uint32_t* out_ids);
cz_status cz_decode_tokens_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens,
                                 const size_t* token_counts, uint32_t context_window, uint32_t reprime_interval,
                                 uint32_t bos, uint32_t* const* out_ids);

/* Segment-sharded encode of ONE stream: like cz_encode_tokens_batch, but only the coded tokens i in [i_begin[k], i_end[k]) of
 * stream k are evaluated, and out_c_lo[k] / out_c_hi[k] (room for i_end[k] - i_begin[k] entries) are indexed by i - i_begin[k].  A re-prime
 * segment of the reference's schedule (src/main.rs:2280-2284) depends on token ids only, so ranks can each take a range of the same file and
 * send their pairs to the rank that runs cz_ac_encode_pairs and writes the single .canz (src/main.rs:2358-2376).  Any range is legal; ranges
 * cut at multiples of reprime_interval cost no duplicated rows.  smollm backend only (RWKV7 has no re-prime: one stream does not shard). */
cz_status cz_encode_tokens_range_batch(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n,
                                       uint32_t context, uint32_t reprime_interval, const size_t* i_begin, const size_t* i_end,
                                       uint32_t* const* out_c_lo, uint32_t* const* out_c_hi);

/* The decode loop of decode_file (src/main.rs:2528-2541, 2621-2628 / 2812-2834) in slices, e.g. to report progress the way the reference's
 * progress bar does (:2354) or to bound the latency of one call: begin copies the payloads and steps BOS; every advance decodes up to
 * n_tokens more tokens per stream in lock-step (*position = tokens decoded so far of the longest stream, *remaining = what is left of it;
 * either may be NULL); fetch copies the ids decoded so far (token_count + 1 per stream, BOS first; entries past the decoded position are
 * unspecified); end releases the slot.  KV cache / recurrent state, arithmetic-decoder states and ids stay in HBM between the calls, so
 * every lock-step group must stay resident: n_streams <= max_streams x lanes (CZ_ERR_UNSUPPORTED otherwise).  cz_decode_tokens_batch is
 * begin + advance(everything) + fetch + end.  One decode in progress per session. */
cz_status cz_decode_begin_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens,
                                const size_t* token_counts, uint32_t context_window, uint32_t reprime_interval, uint32_t bos);
/* Payloads that arrive in pieces (a file still being read, or an encoder still running): begin with the bytes at hand and a capacity per
 * stream, append with feed (final != 0 with the last bytes: only then do reads past the end mean the reference's EOF bits, src/main.rs:432),
 * and advance no further than the bytes fed allow — the decoder looks 32 bits ahead of the symbol it decodes.  A decoder that did read past
 * unfed bytes is reported by cz_decode_fetch_batch (CZ_ERR_STATE), never silently. */
cz_status cz_decode_begin_partial_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens, const size_t* payload_caps,
                                        const size_t* token_counts, uint32_t context_window, uint32_t reprime_interval, uint32_t bos);
cz_status cz_decode_feed_batch(cz_session* s, size_t n_streams, const uint8_t* const* chunks, const size_t* lens, int final_input);
cz_status cz_decode_advance(cz_session* s, size_t n_tokens, size_t* position, size_t* remaining);
cz_status cz_decode_fetch_batch(cz_session* s, size_t n_streams, uint32_t* const* out_ids);
cz_status cz_decode_end(cz_session* s);

/* cross_entropy_bits_over_span (src/main.rs:1725-1751, floor-softmax :758-766), the scorer next to the coder on the same forward:
 * prime = last min(511 - min(n_hint, 511), n_history) ids of history ++ first min(n_hint, 511) ids of hint; bits = sum over the targets of
 * -log2(max(p(target | prime, earlier targets), 1e-300)).  Teacher-forced and batched on the GPU (one causal forward per span, all spans
 * of a batch share the GEMMs); the session's single-stream state is left untouched.  Errors like the reference: empty prime ->
 * CZ_ERR_ARG ("reprime called with empty history"); n_targets == 0 -> 0.0.
 * rwkv7 backend = cross_entropy_bits_over_span_rwkv (src/main.rs:1753-1787): prime = history WITHOUT its literal symbols (id >= vocab) ++ hint,
 * nothing cut (max_context_length() is usize::MAX there, src/models.rs:150); p = combined_pdf_with_literals(logits)[target], ac_p_min()
 * for a target >= vocab + 256; the model steps on vocabulary targets only.  A hint token >= vocab is CZ_ERR_ARG (the reference's embedding
 * lookup fails there). */
cz_status cz_xe_bits_over_span(cz_session* s, const uint32_t* history, size_t n_history, const uint32_t* hint, size_t n_hint,
                               const uint32_t* targets, size_t n_targets, double* bits_out);
cz_status cz_xe_bits_batch(cz_session* s, size_t n_spans, const uint32_t* const* history, const size_t* n_history,
                           const uint32_t* const* hint, const size_t* n_hint, const uint32_t* const* targets, const size_t* n_targets,
                           double* bits_out /* [n_spans] */);

/* ArithmeticEncoder over precomputed intervals (src/main.rs:261-318, 353-399: new + encode_counts* + finish).
 * *payload is malloc'd by the library; release with cz_free.  *zero_width (optional) counts c_lo == c_hi events,
 * i.e. tokens the reference would encode undecodably. */
cz_status cz_ac_encode_pairs(const uint32_t* c_lo, const uint32_t* c_hi, size_t n, uint8_t** payload, size_t* len,
                             uint64_t* zero_width);
void cz_free(void* p);
/* The same encoder fed in pieces (ArithmeticEncoder::encode_counts called window by window, src/main.rs:2300): push returns the bytes that
 * became final with these intervals (*new_bytes points into the encoder, valid until the next call), finish the last ones
 * (src/main.rs:387-399).  The concatenation equals cz_ac_encode_pairs over all the intervals. */
typedef struct cz_ac_encoder cz_ac_encoder;
cz_status cz_ac_encoder_new(cz_ac_encoder** out);
cz_status cz_ac_encoder_push(cz_ac_encoder* e, const uint32_t* c_lo, const uint32_t* c_hi, size_t n, const uint8_t** new_bytes, size_t* new_len);
cz_status cz_ac_encoder_finish(cz_ac_encoder* e, const uint8_t** new_bytes, size_t* new_len, size_t* total_len, uint64_t* zero_width);
void cz_ac_encoder_free(cz_ac_encoder* e);

/* .canz container, header v2 (src/main.rs:551-646) */
typedef struct cz_header {
    uint32_t bos_token_id;
    uint64_t token_count, orig_len_bytes;
    uint8_t model_hash16[16], tokenizer_hash16[16], orig_hash16[16];
    uint32_t reserved_flags, context_window, vocab_size, model_file_repr_len, reprime_interval;
} cz_header;
size_t cz_write_header_v2(const cz_header* h, const uint8_t* model_file_repr, uint8_t* out, size_t cap);   /* :594-611; 0 = cap too small */
cz_status cz_read_header_v2(const uint8_t* buf, size_t len, cz_header* h, size_t* repr_off, size_t* consumed); /* :613-646 */
uint32_t cz_flags_pack(int agent_used, int agent_mock, int gates_present, uint32_t agent_chunk);            /* :247-255 */
/* Producer mark.  The reference leaves bits 1..15 of reserved_flags "reserved (0)" (src/main.rs:240-243; it defines bits 0-2 and reads
 * nothing else, :256-259).  A payload only decodes with the logits arithmetic that encoded it, and this library's arithmetic (canonical
 * chunk-64 f32 order, own expf: DESIGN.md) is not candle's, although the container is the same.  Files written through this library set
 * bit 15, so that a decoder can tell them from candle-produced ones of the same weights: the CLI refuses a file without the mark unless
 * --force-mismatch, next to the model/tokenizer hash checks of src/main.rs:2448-2459.  The reference's reader ignores the bit. */
#define CZ_FLAG_PRODUCER_HIP (1u << 15)
/* gate section after the header when FLAG_AGENT_GATES is set (src/main.rs:2469-2484): returns bytes to skip */
cz_status cz_skip_gate_section(const uint8_t* buf, size_t len, size_t* consumed);

/* BLAKE3 truncated to 16 bytes (src/main.rs:901-939) */
void cz_blake3_bytes_bin16(const uint8_t* p, size_t n, uint8_t out[16]);                                  /* :901-906 */
cz_status cz_blake3_file_bin16(const char* path, uint8_t out[16]);                                        /* :908-921 */
cz_status cz_blake3_files_bin16(const char* const* paths, size_t n, uint8_t out[16]);                     /* :922-939 (hashes file NAMES too) */

/* RWKV "World" tokenizer (candle_rwkv7/src/models/rwkv7.rs:546-610) + symbol planning (src/main.rs:833-895) */
typedef struct cz_rwkv_tokenizer cz_rwkv_tokenizer;
cz_status cz_rwkv_tokenizer_open(const char* vocab_json_path, cz_rwkv_tokenizer** out);
void cz_rwkv_tokenizer_close(cz_rwkv_tokenizer* t);
/* plan_rwkv_symbols: literal escape = vocab + byte.  ids_out has room for ids_cap entries; n_bytes entries always suffice for a
 * vocabulary whose ids are unique (every token covers >= 1 byte).  *n_ids is set even when the call fails with CZ_ERR_ARG because
 * ids_cap is too small (two tokens sharing one id can make the "uncovered tail" of src/main.rs:850-859 longer than n_bytes). */
cz_status cz_rwkv_plan_symbols(const cz_rwkv_tokenizer* t, const uint8_t* bytes, size_t n_bytes, uint32_t vocab,
                               uint32_t* ids_out, size_t ids_cap, size_t* n_ids);
/* rwkv_detok_with_literals: *out malloc'd (cz_free) */
cz_status cz_rwkv_detok(const cz_rwkv_tokenizer* t, const uint32_t* ids, size_t n, uint32_t vocab, uint8_t** out, size_t* out_len);

/* SmolLM2 tokenizer: HF `tokenizers` byte-level BPE read from tokenizer.json (src/main.rs:1812 Tokenizer::from_file).
 * Supported layout: BPE model, no normalizer, ByteLevel (optionally after Digits) pre-tokenizer, ByteLevel decoder; anything
 * else is refused with CZ_ERR_UNSUPPORTED.  Outputs are malloc'd (cz_free). */
typedef struct cz_bpe_tokenizer cz_bpe_tokenizer;
cz_status cz_bpe_tokenizer_open(const char* tokenizer_json_path, cz_bpe_tokenizer** out);
void cz_bpe_tokenizer_close(cz_bpe_tokenizer* t);
size_t cz_bpe_vocab_size(const cz_bpe_tokenizer* t);                                     /* get_vocab_size(true) */
/* tok.encode(String::from_utf8_lossy(data), false)  (src/main.rs:1852): ids without BOS */
cz_status cz_bpe_encode(const cz_bpe_tokenizer* t, const uint8_t* utf8, size_t n, uint32_t** ids_out, size_t* n_ids);
/* tokenizer.decode(ids, skip_special_tokens)  (src/main.rs:2654) */
cz_status cz_bpe_decode(const cz_bpe_tokenizer* t, const uint32_t* ids, size_t n, int skip_special, uint8_t** out, size_t* out_len);

/* anyhow::Error text of the last failing call on this thread */
const char* cz_last_error(void);

/* ---- measurement hooks (bench.py): per-kernel-family HIP-event timing on the library's own stream ---- */
typedef struct cz_kernel_stat {
    char name[48];
    uint64_t launches;
    double total_ms;          /* sum of hipEventElapsedTime over the launches */
    double flops, bytes;      /* algorithmic work attributed to those launches (DESIGN.md "Kernels") */
} cz_kernel_stat;
cz_status cz_profile_enable(cz_session* s, int on);                       /* resets the accumulators */
cz_status cz_profile_read(cz_session* s, cz_kernel_stat* out, size_t cap, size_t* n);
/* wait for the work of THIS session: its own HIP stream, its decode lanes' and its forks' (and their lanes'), each with
 * hipStreamSynchronize — other sessions, torch or RCCL work on the same device are not waited for (bench brackets the timed
 * region with it) */
cz_status cz_synchronize(cz_session* s);

#ifdef __cplusplus
}
#endif
#endif

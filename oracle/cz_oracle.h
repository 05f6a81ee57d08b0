/*
 * cz_oracle.h — CPU ORACLE for the candlezip hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (turtle261/candlezip @ 2025-10-10) for
 * the path  LM forward -> softmax -> 2^30 integer CDF -> 32-bit binary arithmetic coder -> .canz.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (candlezip_amd/, include/) never includes, links or calls anything under oracle/.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - AC / PDF / CDF / container (src/main.rs:227-259, 261-449, 500-677, 758-824): restated line by
 *     line from the reference source; pinned by the reference's committed .canz headers
 *     (tests/golden/) and by hand-derived KATs (the reference has no tests for these functions).
 *   - SmolLM2 (LLaMA) forward: the arithmetic lives in candle-transformers 0.9.1 (not vendored under
 *     /root/reference, Cargo.toml:33-35) -> logits are "PARITY UNPINNED" against candle; the oracle
 *     fixes ONE deterministic f32 evaluation order (canonical chunk-64 fmaf reduction, below) which
 *     the HIP path must reproduce bit for bit.
 *   - RWKV7: restated from candle_rwkv7/src/models/rwkv7.rs (in tree); tensor-op reduction order is
 *     again the canonical one; "PARITY UNPINNED" against candle (no weights / fixtures on disk).
 *
 * Canonical f32 reduction ("cz_dot"): a length-K dot product is evaluated in chunks of 64 along K;
 * inside a chunk   p = fmaf(a[k], b[k], p)  for k ascending starting from p = +0.0f;  the chunk
 * partials are then added left to right in f32:  t = p0; t = t + p1; ...   This is exactly what a
 * chain of v_mfma_f32_* (f32 in, f32 accumulate) instructions computes per 64-deep K tile.
 */
#ifndef CZ_ORACLE_H
#define CZ_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CZO_AC_CDF_TOTAL (1u << 30)     /* src/main.rs:803 */
#define CZO_MAGIC 0x5a505447u           /* "GPTZ"  src/main.rs:227 */
#define CZO_VERSION 2                   /* src/main.rs:228 */

/* ---------------- scalar numerics shared by every model op ---------------- */
float czo_expf(float x);                /* owned expf (H5): only +,*,fma and integer ops */
float czo_logf(float x);
float czo_tanhf(float x);
float czo_dot(const float* a, const float* b, int K);   /* canonical chunk-64 fmaf reduction */
float czo_sum(const float* a, int K);                   /* canonical chunk-64 sequential sum  */

/* ---------------- A-2 / A-3 / R-2: logits -> pdf -> cdf ---------------- */
/* src/main.rs:784-801 */
void czo_softmax_pdf(const float* logits, size_t vocab, double* pdf);
/* src/main.rs:758-767 */
void czo_softmax_pdf_floor(const float* logits, size_t vocab, double p_floor, double* pdf);
/* src/main.rs:769-782 ; pdf has vocab+256 entries */
void czo_combined_pdf_with_literals(const float* logits, size_t vocab, double* pdf);
/* src/main.rs:805-824 ; cdf has n+1 entries */
void czo_quantize_pdf_to_cdf(const double* pdf, size_t n, uint32_t* cdf);
/* convenience: smollm = softmax_pdf + quantize (cdf: vocab+1); rwkv = combined + quantize (cdf: vocab+257) */
void czo_logits_to_cdf_smollm(const float* logits, size_t vocab, uint32_t* cdf);
void czo_logits_to_cdf_rwkv(const float* logits, size_t vocab, uint32_t* cdf);
double czo_ac_p_min(void);              /* src/main.rs:235-238 */

/* ---------------- A-4: arithmetic encoder (src/main.rs:261-318, 353-404) ---------------- */
typedef struct czo_encoder {
    uint64_t low, high, carry_run;
    uint8_t bit_buffer, bit_count;
    uint8_t* out; size_t out_len, out_cap;
    uint64_t zero_width;                /* H3 diagnostics: # of encode_counts calls with c_lo == c_hi */
} czo_encoder;
void czo_enc_init(czo_encoder* e);
void czo_enc_encode_counts(czo_encoder* e, uint64_t c_lo, uint64_t c_hi, uint64_t total);
/* finish(): returns payload length; *out points at encoder-owned memory, valid until czo_enc_free */
size_t czo_enc_finish(czo_encoder* e, const uint8_t** out);
void czo_enc_free(czo_encoder* e);

/* ---------------- A-5: arithmetic decoder (src/main.rs:406-449, 500-548) ---------------- */
typedef struct czo_decoder {
    uint64_t low, high, code;
    const uint8_t* in; size_t in_len, byte_pos; uint8_t bit_pos;
} czo_decoder;
void czo_dec_init(czo_decoder* d, const uint8_t* in, size_t len);
/* cdf has n+1 entries (cdf[0]=0, cdf[n]=total) */
uint32_t czo_dec_decode_symbol_counts(czo_decoder* d, const uint32_t* cdf, size_t n, uint32_t total);

/* ---------------- A-8: container header v2 (src/main.rs:551-646) ---------------- */
typedef struct czo_header {
    uint32_t bos_token_id;
    uint64_t token_count, orig_len_bytes;
    uint8_t model_hash16[16], tokenizer_hash16[16], orig_hash16[16];
    uint32_t reserved_flags, context_window, vocab_size, model_file_repr_len, reprime_interval;
} czo_header;
/* returns bytes written (header + repr) or 0 if cap too small */
size_t czo_write_header_v2(const czo_header* h, const uint8_t* repr, uint8_t* out, size_t cap);
/* returns bytes consumed (header + repr), 0 on error (bad magic / version / truncated);
 * repr_off receives the offset of the model-file name inside buf */
size_t czo_read_header_v2(const uint8_t* buf, size_t len, czo_header* h, size_t* repr_off);
size_t czo_write_var_u64(uint64_t v, uint8_t* out);                     /* src/main.rs:566-575 */
size_t czo_read_var_u64(const uint8_t* in, size_t len, uint64_t* v);    /* src/main.rs:577-592 ; 0 on error */
uint32_t czo_flags_pack(int agent_used, int agent_mock, int gates_present, uint32_t agent_chunk); /* :247-255 */

/* ---------------- A-1 / A-1b: SmolLM (LLaMA) session ---------------- */
typedef struct czo_llama_config {
    int vocab, d_model, n_layers, n_heads, n_kv_heads, d_ffn, max_pos;   /* head_dim fixed = 64 */
    float rope_theta, rms_eps;
} czo_llama_config;
void czo_llama_config_smollm2_135m(czo_llama_config* c);
typedef struct czo_smollm czo_smollm;
/* synthetic weights: bf16-representable values from a counter-based generator (DESIGN.md "Synthetic mode") */
czo_smollm* czo_smollm_new_synthetic(const czo_llama_config* cfg, uint64_t seed, float embed_std);
void czo_smollm_free(czo_smollm* s);
size_t czo_smollm_vocab(const czo_smollm* s);
size_t czo_smollm_index_pos(const czo_smollm* s);                      /* src/models.rs:65 */
/* src/models.rs:92-103 */
int czo_smollm_step_logits(czo_smollm* s, uint32_t token, float* logits);
/* src/models.rs:104-119 */
int czo_smollm_reprime_logits(czo_smollm* s, const uint32_t* hist, size_t n, float* logits);
/* raw bf16 weight access so the tests can hand the SAME tensors to the HIP library */
/* cross_entropy_bits_over_span (src/main.rs:1725-1751) */
int czo_smollm_xe_bits_over_span(czo_smollm* s, const uint32_t* history, size_t n_hist, const uint32_t* hint, size_t n_hint,
                                 const uint32_t* targets, size_t n_t, double* bits_out, double* p_out);
const uint16_t* czo_smollm_tensor_bf16(const czo_smollm* s, const char* name, size_t* numel);
/* batched causal forward at positions 0..n-1, logits for rows [first_row, n) (n - first_row rows x vocab).
 * Used to check "step == batched" (the property segment-parallel encode relies on). */
int czo_smollm_forward_rows(czo_smollm* s, const uint32_t* toks, size_t n, size_t first_row, float* logits);

/* ---------------- A-6 / A-7: the reference's token loops, agent off ---------------- */
/* encode (src/main.rs:1913-1916,1935,1979,2275-2300,2344-2345,2358): ids[0] = BOS, n = #ids incl. BOS.
 * Emits per coded token the interval (c_lo,c_hi) it passed to encode_counts (may be NULL) and the
 * payload. Returns payload length; *payload is malloc'd (caller frees with czo_free). */
size_t czo_smollm_encode_tokens(czo_smollm* s, const uint32_t* ids, size_t n, uint32_t context,
                                uint32_t reprime_interval, uint32_t* c_lo, uint32_t* c_hi,
                                uint8_t** payload, uint64_t* zero_width);
/* decode (src/main.rs:2485,2506,2528-2541,2621-2628): out_ids gets token_count+1 entries (BOS first) */
int czo_smollm_decode_tokens(czo_smollm* s, const uint8_t* payload, size_t len, size_t token_count,
                             uint32_t context_window, uint32_t reprime_interval, uint32_t bos,
                             uint32_t* out_ids);
void czo_free(void* p);

/* ---------------- R-1: RWKV7 session ---------------- */
typedef struct czo_rwkv7_config {
    int vocab, hidden, n_layers, head_dim, d_ffn, a_rank, decay_rank, v_rank, gate_rank;
    float norm_eps;
} czo_rwkv7_config;
void czo_rwkv7_config_0b1(czo_rwkv7_config* c);
typedef struct czo_rwkv7 czo_rwkv7;
czo_rwkv7* czo_rwkv7_new_synthetic(const czo_rwkv7_config* cfg, uint64_t seed, float embed_std);
void czo_rwkv7_free(czo_rwkv7* s);
size_t czo_rwkv7_vocab(const czo_rwkv7* s);
int czo_rwkv7_step_logits(czo_rwkv7* s, uint32_t token, float* logits);   /* src/models.rs:150-160 */
void czo_rwkv7_reset(czo_rwkv7* s);
const uint16_t* czo_rwkv7_tensor_bf16(const czo_rwkv7* s, const char* name, size_t* numel);
size_t czo_rwkv7_encode_tokens(czo_rwkv7* s, const uint32_t* ids, size_t n, uint32_t* c_lo, uint32_t* c_hi,
                               uint8_t** payload, uint64_t* zero_width);
int czo_rwkv7_decode_tokens(czo_rwkv7* s, const uint8_t* payload, size_t len, size_t token_count,
                            uint32_t bos, uint32_t* out_ids);
int czo_rwkv7_xe_bits_over_span(czo_rwkv7* s, const uint32_t* history, size_t n_hist, const uint32_t* hint, size_t n_hint,
                                const uint32_t* targets, size_t n_t, double* bits_out, double* p_out);   /* src/main.rs:1753-1787 */

/* synthetic generator primitive (exposed for tests): bf16 bits of element idx of tensor tid */
uint16_t czo_synth_bf16(uint64_t seed, uint32_t tid, uint64_t idx, float std, float mean);

/* ---------------- f1: BLAKE3-128 as the container uses it (src/main.rs:901-939) ---------------- */
typedef struct czo_blake3 czo_blake3;
czo_blake3* czo_blake3_new(void);
void czo_blake3_update(czo_blake3* h, const uint8_t* p, size_t n);
void czo_blake3_final16(czo_blake3* h, uint8_t out[16]);
void czo_blake3_free(czo_blake3* h);
void czo_blake3_bytes_bin16(const uint8_t* p, size_t n, uint8_t out[16]);

/* per-op entry points for kernel-level parity tests */
void czo_gemm_bf16(const float* A, const uint16_t* W, const float* R, float* C, int M, int N, int K);
void czo_rmsnorm_bf16w(const float* X, const uint16_t* w, float* Y, int M, int d, float eps);
void czo_attention_ref(const float* Q, const float* K, const float* V, float* O, int M, int pos0, int nh, int nkv);
void czo_swiglu(const float* GU, float* H, int M, int f);

int czo_num_threads(void);
void czo_set_num_threads(int n);   /* the OpenMP runtime may already be initialised by the host process (torch): env vars are then too late */

#ifdef __cplusplus
}
#endif
#endif

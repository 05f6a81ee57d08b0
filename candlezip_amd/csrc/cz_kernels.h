// cz_kernels.h — launchers of the HIP kernels (internal to libcandlezip_hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cz_numerics.h"

#define CZ_KERNEL_MAX_CTX 8192     // longest context the attention kernels take (max_position_embeddings of SmolLM2-135M)

// one attention "sequence": rows [row0, row0+nrows) of the activation batch sit at positions pos0.. and attend to
// keys/values k[j*kv_stride + kvh*64 + d], j in [0, pos0 + local_row]
struct cz_seq_desc {
    int row0, nrows, pos0, kv_stride;
    const float* k;
    const float* v;
    int kt_stride;      // 0: K rows like V (k[j*kv_stride + kvh*64 + d]); > 0: K transposed, k[(kvh*64 + d)*kt_stride + j] (the decode KV cache:
    int pad_;           //    lane = key then reads 64 consecutive floats per d -> fully coalesced)
};

// rows [row0, row0+nrows) are RoPE'd at positions pos0..; K (RoPE'd) and V are written to k_dst/v_dst rows pos0..
struct cz_rope_desc {
    int row0, nrows, pos0, dst_stride;
    float* k_dst;
    float* v_dst;
    int kt_stride;      // as in cz_seq_desc
    int pad_;
};

// dynamic-LDS limits of the GEMM kernels: call once per device (session open) before the first launch
hipError_t cz_gemm_prepare();
hipError_t cz_gemm_mid_prepare();
hipError_t cz_dec_gemm_prepare();
hipError_t cz_launch_gu_interleave(hipStream_t st, const uint16_t* W, uint16_t* Wi, int f, int K);
hipError_t cz_launch_gemm_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wi, float* H, int ldh, int M, int f, int K);
hipError_t cz_launch_gemm(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int ldr,
                          int M, int N, int K, int force_big);
hipError_t cz_launch_gemm_masked(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K, const unsigned char* row_mask);
hipError_t cz_launch_gemm_relusq(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K);
// mid-M GEMM for one wide lock-step decode group (kernels_gemm_mid.hip); Wk = k2-packed copy of the weights (cz_launch_k2_pack)
hipError_t cz_launch_k2_pack(hipStream_t st, const uint16_t* W, uint16_t* Wk, size_t N, int K, int rot_heads);   // rot_heads: leading 64-row heads stored pair-wise for the RoPE epilogue
hipError_t cz_launch_gemm_mid(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, const float* R, int ldr, int M, int N, int K);
hipError_t cz_launch_gemm_mid_relusq(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, int M, int N, int K);
hipError_t cz_launch_gemm_mid_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* Wk, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M);
hipError_t cz_launch_gemm_mid_qkv(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn);
hipError_t cz_launch_gemm_mid_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* H, int ldh, int M, int f, int K);
hipError_t cz_launch_embed(hipStream_t st, const uint32_t* tok, const uint16_t* E, float* X, int M, int d);
hipError_t cz_launch_rmsnorm(hipStream_t st, const float* X, int ldx, const uint16_t* w, float* Y, int ldy, int M, int d, float eps,
                             const int* row_index);
hipError_t cz_launch_gather_rows(hipStream_t st, const float* X, int ldx, const int* rows, float* Y, int ldy, int M, int d);
hipError_t cz_launch_rope(hipStream_t st, float* QKV, int ld, const cz_rope_desc* descs, int nseq, int max_rows, int n_heads,
                          int n_kv, const float* cs, const float* sn, size_t kv_off, const int* pos_dev);
hipError_t cz_launch_swiglu(hipStream_t st, const float* GU, float* H, int M, int f);
hipError_t cz_launch_attention(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nseq,
                               int max_rows, int max_nk, int n_heads, int n_kv, size_t kv_off, const int* pos_dev);
size_t cz_attention_smem_bytes(int max_nk);
hipError_t cz_attention_prepare(int max_nk);

// logits -> coder interval of `sym` (encode) ; mode 0 = smollm (softmax_pdf), 1 = rwkv7 (floor + 256 literals)
// (exact-parallel kernel when supported, else the serial walk; fallback_count (optional, device) counts rows that fell back)
hipError_t cz_launch_cdf_pairs(hipStream_t st, const float* logits, int ld, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi,
                               int rows, int vocab, int mode, unsigned* fallback_count, double* scratch,
                               const unsigned long long* out_step_dev = nullptr);   // non-null: pairs go to c_lo/c_hi + *out_step_dev * rows
// floor-softmax probability of sym[row] (f64) for the cross-entropy span scorer (src/main.rs:1725-1751, 758-766)
// mode 0: softmax_pdf_floor(logits)[sym] (the generic cross_entropy_bits_over_span); mode 1: combined_pdf_with_literals(logits)[sym] (..._rwkv)
hipError_t cz_launch_xe_probs(hipStream_t st, const float* logits, int ld, const uint32_t* sym, double* p_out, int rows, int vocab, unsigned* fallback_count,
                              double* scratch, int mode = 0);
#define CZ_CDF_PAR_MAX_GRID 1024
size_t cz_cdf_par_scratch_bytes(int vocab, int mode);     // the `scratch` the exact-parallel CDF launchers need (owned by the session)
hipError_t cz_launch_cdf_pairs_serial(hipStream_t st, const float* logits, int ld, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi,
                                      int rows, int vocab, int mode);
bool cz_cdf_par_supported(int vocab, int mode);
// device-resident arithmetic decoder state, one per stream (src/main.rs:406-417)
struct cz_dec_state {
    unsigned long long low, high, code;
    unsigned long long byte_pos;
    unsigned int bit_pos, active;
    const uint8_t* in;
    unsigned long long in_len;
    uint32_t* out_ids;            // out_ids[0] = bos, out_ids[i+1] = i-th decoded symbol
    unsigned long long token_count;
    unsigned int final_input;     // 1: `in_len` bytes are the whole payload (reads past them are the reference's EOF bits = 1); 0: more bytes will be fed
    unsigned int underrun;        // set when the decoder read past the bytes fed so far (final_input == 0): its ids are then garbage
};
hipError_t cz_launch_dec_feed(hipStream_t st, cz_dec_state* st_arr, int nstreams, const unsigned long long* new_len_dev, int final_input);
hipError_t cz_launch_dec_init(hipStream_t st, cz_dec_state* st_arr, int nstreams);
hipError_t cz_launch_decode_symbol(hipStream_t st, const float* logits, int ld, cz_dec_state* st_arr, uint32_t* sym_out, int nstreams,
                                   int vocab, int mode, unsigned long long i, unsigned* fallback_count, const unsigned long long* step_dev,
                                   double* scratch);
hipError_t cz_launch_decode_symbol_serial(hipStream_t st, const float* logits, int ld, cz_dec_state* st_arr, uint32_t* sym_out, int nstreams,
                                          int vocab, int mode, unsigned long long i, const unsigned long long* step_dev);

// ---- lock-step decode path (kernels_decode.hip); "kq" = k-interleaved layout, see the file header
hipError_t cz_launch_dec_pack(hipStream_t st, const uint16_t* W, uint16_t* Wd, size_t N, int K);
hipError_t cz_launch_dec_rmsnorm(hipStream_t st, const float* X, int ldx, const uint16_t* w, float* Y, int ldy, int M, int d, float eps);
// normw != nullptr: A is the residual stream in natural layout and RMSNorm(normw, eps) is applied on the fly
hipError_t cz_launch_dec_gemm_plain(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, const float* R, int M, int N, int K,
                                    const uint16_t* normw, float eps);
hipError_t cz_launch_dec_gemm_qkv(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn,
                                  const uint16_t* normw, float eps);
hipError_t cz_launch_dec_gemm_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* H, int ldh, int M, int f, int K,
                                     const uint16_t* normw, float eps);
bool cz_dec_gemm_norm_is_cheap(int K);
hipError_t cz_launch_dec_gemm_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* Wd, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M);
hipError_t cz_launch_dec_gemm_masked(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, const float* R, int M, int N, int K,
                                     const unsigned char* row_mask);
hipError_t cz_launch_dec_gemm_relusq(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, int M, int N, int K);
// the same projections for a handful of streams (<= 4 rows) on the VALU, one wave per block (kernels_gemv.hip); A and W in NATURAL layout
bool cz_dec_gemv_ok(int M, int K, bool norm);
hipError_t cz_launch_dec_gemv_plain(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int M, int N, int K,
                                    const uint16_t* normw, float eps);
hipError_t cz_launch_dec_gemv_qkv(hipStream_t st, const float* A, int lda, const uint16_t* W, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn, const uint16_t* normw, float eps);
hipError_t cz_launch_dec_gemv_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* W, float* H, int ldh, int M, int f, int K, const uint16_t* normw, float eps);
hipError_t cz_launch_dec_gemv_relusq(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K);
hipError_t cz_launch_dec_gemv_masked(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K, const unsigned char* row_mask);
hipError_t cz_launch_dec_gemv_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* W, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M);
hipError_t cz_dec_attn_prepare(int max_nk);
// split_scratch (cz_dec_attn_split_bytes(split_streams, n_heads) bytes, or null): scores / chunk maxima / P.V partials of the three-launch form
// small groups take (kernels_decode.hip, "Split decode attention")
hipError_t cz_launch_dec_attn(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nstreams, int max_nk,
                              int n_heads, int n_kv, size_t kv_off, const int* pos_dev, int out_kq = 1, float* split_scratch = nullptr, int split_streams = 0, int concurrent_lanes = 1);
size_t cz_dec_attn_split_bytes(int nstreams, int n_heads);
bool cz_dec_attn_is_split(int nstreams, int max_nk, int n_heads, int n_kv, const float* split_scratch, int split_streams);
const float* cz_dec_attn_split_partials(const float* split_scratch, int split_streams, int n_heads, int* maxc);
hipError_t cz_launch_dec_gemv_attsum(hipStream_t st, const float* part, int part_maxc, int max_nk, const int* pos_dev, const uint16_t* W, float* C, int ldc, const float* R,
                                     int M, int N, int n_heads);

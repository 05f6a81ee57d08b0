// cz_session.h — the session object behind the opaque cz_session handle (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include "../../include/candlezip.h"
#include "cz_kernels.h"

#define CZ_SYNC_EVERY 8      // decode loops wait for the device every this many tokens (bounds the packets queued behind the host; session.hip)
#define CZ_DEF_CTX 2048      // positions per stream the KV cache starts with (the reference's default sawtooth peaks at 1023, config 5's at 1535);
                             // a run that needs more (a re-prime interval > 1536) gets a cache sized for it, up to cz_ctx_limit()

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    unsigned long long* gen = nullptr;   // bumped when the buffer moves (captured graphs hold raw pointers)
    cz_status ensure(size_t bytes);
    void release();
};

enum ProfKind { PK_GEMM = 0, PK_GEMM_SKINNY, PK_ATTN, PK_RMSNORM, PK_ROPE, PK_SWIGLU, PK_EMBED, PK_CDF, PK_DECODE, PK_RWKV, PK_DEC_GEMM, PK_DEC_ATTN, PK_DEC_NORM, PK_DEC_GEMM_QKV, PK_DEC_GEMM_GU, PK_DEC_GEMM_HEAD, PK_GEMM_N64, PK_RW_STATE, PK_RW_PREP, PK_RW_GEMM, PK_RW_SCAN, PK_COUNT };
struct ProfAcc { uint64_t launches = 0; double ms = 0, flops = 0, bytes = 0; };
struct ProfEvt { hipEvent_t e0, e1; int kind; double flops, bytes; };

struct LlamaModel {
    cz_llama_config cfg{};
    uint16_t* blob = nullptr;          // one allocation: E | per-layer tensors | lnf   (bf16)
    uint16_t* blob_dec = nullptr;      // the same tensors with K in kq order (decode path); same offsets as blob
    uint16_t* E = nullptr;             // [vocab][d]  (tied lm_head)
    struct Layer { uint16_t *ln1, *wqkv, *wo, *ln2, *wgu, *wd, *wgu_i; };   // wgu_i: gate|up rows interleaved by 32 for the fused SwiGLU epilogue (big-M path), or null
    uint16_t* blob_gu = nullptr;       // storage of the wgu_i copies
    uint16_t* blob_k2 = nullptr;       // the layer matrices packed for the mid-M GEMM (kernels_gemm_mid.hip); wgu's slot holds the interleaved gate|up rows
    std::vector<Layer> L;
    uint16_t* lnf = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;   // [max_pos][32]
    size_t param_count = 0;
};

// f32 bits -> f32 bits of the nearest-even bf16 (NaN stays NaN)
static inline uint32_t cz_f32_bits_round_bf16(uint32_t u) {
    if ((u & 0x7f800000u) == 0x7f800000u) return u & 0xffff0000u ? (u | 0x00400000u) & 0xffff0000u : u & 0xffff0000u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}

struct Rwkv7Model;   // rwkv7.hip

// a lock-step decode in progress on one lane (cz_decode_begin_batch ... cz_decode_end): host-side cursor of the token loop; everything
// else (KV cache / recurrent state, arithmetic-decoder states, decoded ids, position counters) is device-resident
struct DecRun {
    bool active = false, use_graph = false;
    int B = 0, max_nk = 0;
    std::vector<size_t> tcs;
    size_t max_tc = 0, i = 0, index_pos = 0, ec = 0, ids_total = 0;
    uint32_t R = 0, bos = 0;
    bool partial = false;              // payloads arrive in pieces (cz_decode_begin_partial_batch / cz_decode_feed_batch)
    std::vector<size_t> pay_off, pay_len, pay_cap;
};

struct cz_session {
    cz_backend backend = CZ_BACKEND_SMOLLM;
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t seed = 0;
    bool round_f32 = false;            // cz_model_desc.round_f32_weights_to_bf16
    LlamaModel llama;
    Rwkv7Model* rwkv = nullptr;
    // packed-row work buffers
    DevBuf tok, x, xn, qkv, gu, h, seqd, roped, rowidx, sym, clo, chi, xl, logits, payload, outids, decst, seqd_q, xc, rc;   // seqd_q, xc, rc: llama_body's last layer on the wanted rows only
    size_t max_batch_rows = 98304, logit_batch_rows = 8192;
    int force_big_gemm = 0, serial_cdf = 0, decode_v1 = 0, no_graph = 0, fuse_norm = 0, step_sync = 0, dbg_skip = 0, wide_min_streams = 144, wide_mode = 3;     // (wide_min_streams: 144 streams 2.90 ms per step on the wide form against 3.27 on the narrow one, 160: 3.00 / 3.40, 128: 2.78 / 2.62 — profiles/r04_single_stream/steps_b20_to_b256_step_forms.jsonl)
    hipGraph_t dbg_g = nullptr; hipGraphExec_t dbg_ge = nullptr;      // cz_debug_decode_steps
    int rwkv_wide_mode = 1;             // rw_step_wide's projections: 1 = gemm_mid_kernel (grouped launches), 0 = the tiled big-M kernel
    int rwkv_wide_min = 80;             // RWKV7 lock-step groups of at least this many streams take the wide step (rw_step_wide: grouped gemm_mid launches); 0 = never.  Round 4, decompress
                                        // tokens/s narrow / wide: 64 streams 52.1 k / 50.1 k, 96: 66.2 k / 69.9 k, 128: 84.2 k / 88.5 k, 160: 85.8 k / 101.3 k, 192: 86.9 k / 114.7 k (profiles/r04_single_stream/)
    int rwkv_seq = 1;                   // 1 = RWKV7 encode is sequence-parallel (rw_encode_seq_group); 0 = the lock-step token-by-token walk (cz_rwkv7_encode_group)
    int no_last_prune = 0;              // test knob: 1 = the last layer's tail on every row (llama_body)
    int stream_prio = 0;                // HIP priority of this session's stream and of the decode lanes it makes (cz_debug_set "stream_priority")
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr; int graph_B = 0, graph_max_nk = 0; unsigned long long graph_gen = 0;
    unsigned long long devbuf_gen = 0, graph_launch_ns = 0, graph_launch_cpu_ns = 0, graph_launch_idle_ns = 0; int probe_launch = 0;
    unsigned* fallback_dev = nullptr;   // rows the exact-parallel CDF kernel handed to the serial walk
    double* cdf_scratch = nullptr;      // spill rows of the exact-parallel CDF kernel (cz_cdf_par_scratch_bytes)
    // decode lanes: independent lock-step groups replayed concurrently on their own HIP streams (share the weights)
    std::vector<cz_session*> lanes; bool is_lane = false; int max_lanes = 4;
    int lanes_concurrent = 1;      // lock-step groups of the running call that decode side by side on this device (set by the lane runners; picks the decode attention's block shape)
    cz_session* fork_of = nullptr; std::vector<cz_session*> forks; bool orphan = false;   // cz_session_fork: a fork shares the parent's weights; closing the parent first orphans it (every call then fails)
    std::vector<DevBuf*> all_bufs() {   // ONE list for the constructor and cz_session_close (a buffer missing from either leaks or dangles)
        return {&tok, &x, &xn, &qkv, &gu, &h, &seqd, &roped, &rowidx, &sym, &clo, &chi, &xl, &logits, &payload, &outids, &decst, &seqd_q, &xc, &rc};
    }
    cz_session() { for (DevBuf* b : all_bufs()) b->gen = &devbuf_gen; }
    // lock-step streams
    int max_streams = 64;
    size_t ctx_cap = 0, kv_layer_stride = 0, kv_stream_stride = 0; int kv_ctx_cap = 0;   // kv_ctx_cap: positions per stream the cache starts with (0 = CZ_DEF_CTX)
    float* kvcache = nullptr;          // [stream][layer][K|V][ctx_cap][n_kv*64]
    int* pos_dev = nullptr;
    cz_seq_desc* step_seqd = nullptr; cz_rope_desc* step_roped = nullptr; int* step_rows = nullptr;
    float* step_logits = nullptr; uint32_t* step_sym = nullptr;
    float* attn_split = nullptr; int attn_split_streams = 0;      // scratch of the three-launch decode attention of small groups (cz_launch_dec_attn)
    size_t index_pos = 0;              // SmolLmSession::index_pos (src/models.rs:65)
    DecRun run;                        // this lane's group of a sliced decode
    std::vector<size_t> run_off;       // parent only: group g of the sliced decode = streams [run_off[g], run_off[g+1]) on lane g
    // profiling
    bool prof_on = false, prof_cur = false; unsigned prof_mask = 0xFFFFFFFFu;
    ProfAcc acc[PK_COUNT];
    std::vector<ProfEvt> evts; std::vector<hipEvent_t> evt_free; ProfEvt evt_open{};
    void prof_begin(int kind);
    void prof_end(double flops, double bytes);
    void prof_drain();
};

// rwkv7.hip
cz_status cz_rwkv7_build(cz_session* s, const cz_model_desc* desc);
void cz_rwkv7_destroy(cz_session* s);
size_t cz_rwkv7_vocab(const cz_session* s);
cz_status cz_rwkv7_step_host(cz_session* s, uint32_t token, float* out_logits);
cz_status cz_rwkv7_reprime_host(cz_session* s, const uint32_t* hist, size_t n, float* out_logits);
cz_status cz_rwkv7_make_lane(const cz_session* parent, cz_session* child);
cz_status cz_rwkv7_encode_seq(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n, uint32_t* const* lo, uint32_t* const* hi);
cz_status cz_rwkv7_xe_bits(cz_session* s, size_t n_spans, const uint32_t* const* hist, const size_t* n_hist, const uint32_t* const* hint, const size_t* n_hint,
                           const uint32_t* const* targets, const size_t* n_targets, double* bits_out);
cz_status cz_rwkv7_encode_group(cz_session* s, int B, const uint32_t* const* ids, const size_t* n, uint32_t* const* lo, uint32_t* const* hi);
cz_status cz_rwkv7_decode_begin(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* tcs, uint32_t bos);
// session.hip: payload buffer / decoder states of a lock-step group, shared by both back-ends
cz_status cz_dec_run_setup(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* tcs);
cz_status cz_dec_run_feed(cz_session* s, const uint8_t* const* chunks, const size_t* lens, int final_input);
cz_status cz_dec_run_fetch(cz_session* s, uint32_t* const* out_ids);
cz_status cz_rwkv7_decode_advance(cz_session* s, size_t n);
cz_status cz_rwkv7_decode_group(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* tcs, uint32_t bos,
                                uint32_t* const* out_ids);

// session.hip — model sessions, segment-parallel encode, lock-step decode and the C ABI (include/candlezip.h).
//
// Mirrors the reference's session object and token loops:
//   SmolLmSession (src/models.rs:37-120)            -> cz_session + cz_step_logits / cz_reprime_logits
//   encode_file loop  (src/main.rs:1913-1916, 1935, 1979, 2275-2300, 2344-2345)   -> cz_encode_tokens[_batch]
//   decode_file loop  (src/main.rs:2485, 2506, 2528-2541, 2621-2628)              -> cz_decode_tokens[_batch]
// Design: encode is teacher-forced, so every re-prime segment is an independent causal forward; all
// segments of all streams are packed row-wise into shared f32-MFMA GEMMs.  Decode steps B independent streams in
// lock-step; the arithmetic-decoder state, the decoded ids and the position counter live in HBM, the host only
// enqueues kernels (no per-token synchronisation).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <time.h>
#include <string>
#include "cz_host.h"
#include "cz_kernels.h"
#include "cz_session.h"

extern int g_gemm_lds_kb;
extern int g_attn_v1;
extern int g_dec_grid_target;
extern int g_dec_nw_cap;
extern int g_dec_mt_max;
extern int g_dec_norm_mfma;
extern int g_gemm_mid_rows;
extern int g_mid_nw;
extern int g_mid_colmaj;
extern int g_mid_quad;
extern int g_dec_attn_split_heads;
extern int g_dec_attn_lds_pad;
extern int g_dec_wide_rows;
extern int g_dec_attn_small_grid;
extern int g_dec_attn_split_max;
extern int g_rw_seq_rows;
extern int g_rw_state_pad;
extern int g_gemv_fold_attn;
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            cz_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            return CZ_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)
#define CZCHK(expr) do { cz_status _s = (expr); if (_s != CZ_OK) return _s; } while (0)

// ------------------------------------------------------------------ small device helpers
__global__ void synth_kernel(uint16_t* dst, size_t numel, uint64_t seed, uint32_t tid, float stdv, float mean) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = cz_synth_bf16(seed, tid, i, stdv, mean);
}
__global__ void set_int_kernel(int* p, int v) { *p = v; }
// device-resident counters of the lock-step decode loop: pos_dev[0] = position, (u64*)(pos_dev + 2) = token index i
__global__ void set_counters_kernel(int* p, int pos, unsigned long long step) { p[0] = pos; *reinterpret_cast<unsigned long long*>(p + 2) = step; }
__global__ void inc_counters_kernel(int* p) { p[0] += 1; *reinterpret_cast<unsigned long long*>(p + 2) += 1ull; }
__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void init_out_ids_kernel(cz_dec_state* st, int n, uint32_t bos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st[i].out_ids[0] = bos;
}
// re-prime history of every stream: tok[b*hl + j] = out_ids_b[end - hl + j]  (src/main.rs:2533-2535)
__global__ void gather_hist_kernel(const cz_dec_state* st, uint32_t* tok, int nstreams, int hl, unsigned long long end) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstreams * hl) return;
    const int b = i / hl, j = i % hl;
    unsigned long long idx = end - hl + j;
    const unsigned long long last = st[b].token_count;        // finished streams: clamp inside their buffer
    if (idx > last) idx = last;
    tok[i] = st[b].out_ids[idx];
}

cz_status DevBuf::ensure(size_t bytes) {
    if (bytes <= cap) return CZ_OK;
    if (gen) ++*gen;
    if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    return CZ_OK;
}
void DevBuf::release() { if (p) { hipError_t e = hipFree(p); (void)e; } p = nullptr; cap = 0; }

// ------------------------------------------------------------------ profiling (HIP events on the session stream)
void cz_session::prof_begin(int kind) {
    prof_cur = prof_on && ((prof_mask >> kind) & 1u);
    if (!prof_cur) return;
    ProfEvt ev; ev.kind = kind; ev.flops = 0; ev.bytes = 0;
    if (!evt_free.empty()) { ev.e0 = evt_free.back(); evt_free.pop_back(); ev.e1 = evt_free.back(); evt_free.pop_back(); }
    else { hipEventCreate(&ev.e0); hipEventCreate(&ev.e1); }
    hipEventRecord(ev.e0, stream);
    evt_open = ev;
}
void cz_session::prof_end(double flops, double bytes) {
    if (!prof_cur) return;
    hipEventRecord(evt_open.e1, stream);
    evt_open.flops = flops; evt_open.bytes = bytes;
    evts.push_back(evt_open);
    if (evts.size() >= 8192) prof_drain();
}
void cz_session::prof_drain() {
    if (evts.empty()) return;
    hipStreamSynchronize(stream);
    for (auto& ev : evts) {
        float ms = 0; hipEventElapsedTime(&ms, ev.e0, ev.e1);
        ProfAcc& a = acc[ev.kind];
        a.launches += 1; a.ms += ms; a.flops += ev.flops; a.bytes += ev.bytes;
        evt_free.push_back(ev.e0); evt_free.push_back(ev.e1);
    }
    evts.clear();
}
static const char* kProfNames[PK_COUNT] = {"gemm_bf16w_f32_mfma", "gemm_skinny_f32_mfma", "attention_f32_mfma", "rmsnorm", "rope",
                                           "swiglu", "embed", "cdf_pairs", "decode_symbol", "rwkv7_step", "dec_gemm_plain", "dec_attention", "dec_rmsnorm",
                                           "dec_gemm_qkv_rope", "dec_gemm_swiglu", "dec_gemm_lm_head", "gemm_bf16w_f32_mfma_n64",
                                           "rwkv7_state", "rwkv7_prep", "rwkv7_gemm", "rwkv7_scan"};

// ------------------------------------------------------------------ model construction
static cz_status synth(cz_session* s, uint16_t* dst, const char* name, size_t numel, float stdv, float mean) {
    hipLaunchKernelGGL(synth_kernel, dim3(1024), dim3(256), 0, s->stream, dst, numel, s->seed, cz_fnv1a32(name), stdv, mean);
    HIPCHK(hipGetLastError());
    return CZ_OK;
}
// copy a checkpoint tensor to the device as bf16 (BF16 verbatim; F32/F16 only when every value is bf16-exact)
static cz_status load_tensor(cz_session* s, std::vector<std::unique_ptr<SafeTensorsFile>>& files, const std::string& name,
                             uint16_t* dst, size_t numel) {
    for (auto& f : files) {
        auto it = f->tensors.find(name);
        if (it == f->tensors.end()) continue;
        const SafeTensor& t = it->second;
        size_t n = 1; for (auto d : t.shape) n *= (size_t)d;
        if (n != numel) { cz_set_error("tensor %s has %zu elements, expected %zu", name.c_str(), n, numel); return CZ_ERR_IO; }
        if (t.dtype == "BF16") {
            if (t.nbytes != numel * 2) { cz_set_error("tensor %s: byte size mismatch", name.c_str()); return CZ_ERR_IO; }
            HIPCHK(hipMemcpy(dst, t.data, numel * 2, hipMemcpyHostToDevice));
            return CZ_OK;
        }
        if (t.dtype == "F32") {
            if (t.nbytes != numel * 4) { cz_set_error("tensor %s: byte size mismatch", name.c_str()); return CZ_ERR_IO; }
            std::vector<uint16_t> tmp(numel);
            const uint32_t* src = (const uint32_t*)t.data;
            for (size_t i = 0; i < numel; ++i) {
                uint32_t u; memcpy(&u, src + i, 4);
                if ((u & 0xffffu) && !s->round_f32) { cz_set_error("tensor %s is F32 with values that are not bf16-representable (unsupported; see cz_model_desc.round_f32_weights_to_bf16)", name.c_str()); return CZ_ERR_UNSUPPORTED; }
                if (u & 0xffffu) u = cz_f32_bits_round_bf16(u);
                tmp[i] = (uint16_t)(u >> 16);
            }
            HIPCHK(hipMemcpy(dst, tmp.data(), numel * 2, hipMemcpyHostToDevice));
            return CZ_OK;
        }
        cz_set_error("tensor %s has unsupported dtype %s", name.c_str(), t.dtype.c_str());
        return CZ_ERR_UNSUPPORTED;
    }
    cz_set_error("tensor %s not found in the safetensors files", name.c_str());
    return CZ_ERR_IO;
}

static cz_status read_file(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) { cz_set_error("failed reading %s", path); return CZ_ERR_IO; }
    char buf[1 << 14]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f); return CZ_OK;
}
static cz_status parse_llama_config(const char* path, cz_llama_config* c) {     // HF config.json -> LlamaConfig (src/models.rs:49-53)
    std::string txt; CZCHK(read_file(path, &txt));
    JsonValue root; std::string err;
    if (!cz_json_parse(txt.data(), txt.size(), &root, &err)) { cz_set_error("failed to parse config.json as HF LlamaConfig: %s", err.c_str()); return CZ_ERR_IO; }
    auto num = [&](const char* k, double def) { const JsonValue* v = root.get(k); return v && v->kind == JsonValue::Num ? v->num : def; };
    c->vocab = (int)num("vocab_size", 0); c->d_model = (int)num("hidden_size", 0); c->n_layers = (int)num("num_hidden_layers", 0);
    c->n_heads = (int)num("num_attention_heads", 0); c->n_kv_heads = (int)num("num_key_value_heads", c->n_heads);
    c->d_ffn = (int)num("intermediate_size", 0); c->max_pos = (int)num("max_position_embeddings", 2048);
    c->rope_theta = (float)num("rope_theta", 10000.0); c->rms_eps = (float)num("rms_norm_eps", 1e-5);
    return CZ_OK;
}

static cz_status build_llama(cz_session* s, const cz_model_desc* desc) {
    LlamaModel& m = s->llama;
    if (desc->config_path) CZCHK(parse_llama_config(desc->config_path, &m.cfg)); else m.cfg = desc->llama;
    const cz_llama_config& c = m.cfg;
    if (c.n_heads <= 0 || c.n_kv_heads <= 0 || c.d_model != c.n_heads * 64 || c.n_heads % c.n_kv_heads || 16 / (c.n_heads / c.n_kv_heads) < 1 ||
        c.d_model % 64 || c.d_ffn % 64 || c.vocab % 64 || c.d_model > 1024 || c.n_layers <= 0 || (c.d_model + 2 * c.n_kv_heads * 64) % 64) {
        cz_set_error("unsupported LLaMA shape (need head_dim 64, d_model/d_ffn/vocab multiples of 64, d_model <= 1024)");
        return CZ_ERR_UNSUPPORTED;
    }
    const size_t d = c.d_model, kvd = (size_t)c.n_kv_heads * 64, f = c.d_ffn, V = c.vocab, L = c.n_layers;
    const size_t per_layer = d + (d + 2 * kvd) * d + d * d + d + 2 * f * d + d * f;
    const size_t total = V * d + L * per_layer + d;
    HIPCHK(hipMalloc((void**)&m.blob, total * 2));
    uint16_t* p = m.blob;
    m.E = p; p += V * d;
    m.L.resize(L);
    for (size_t l = 0; l < L; ++l) {
        LlamaModel::Layer& y = m.L[l];
        y.ln1 = p; p += d; y.wqkv = p; p += (d + 2 * kvd) * d; y.wo = p; p += d * d; y.ln2 = p; p += d;
        y.wgu = p; p += 2 * f * d; y.wd = p; p += d * f;
    }
    m.lnf = p; p += d;
    m.param_count = V * d + L * ((d + 2 * kvd) * d + d * d + 2 * f * d + d * f);
    std::vector<std::unique_ptr<SafeTensorsFile>> files;
    const bool synthetic = desc->n_weight_paths == 0;
    for (size_t i = 0; i < desc->n_weight_paths; ++i) {
        files.emplace_back(new SafeTensorsFile());
        if (!cz_safetensors_open(desc->weight_paths[i], files.back().get())) return CZ_ERR_IO;
    }
    auto get = [&](const std::string& name, uint16_t* dst, size_t numel, float stdv, float mean) -> cz_status {
        return synthetic ? synth(s, dst, name.c_str(), numel, stdv, mean) : load_tensor(s, files, name, dst, numel);
    };
    CZCHK(get("model.embed_tokens.weight", m.E, V * d, desc->synthetic_embed_std, 0.0f));   // lm_head tied (tie_word_embeddings)
    for (size_t l = 0; l < L; ++l) {
        const std::string pre = "model.layers." + std::to_string(l) + ".";
        LlamaModel::Layer& y = m.L[l];
        CZCHK(get(pre + "input_layernorm.weight", y.ln1, d, 0.1f, 1.0f));
        CZCHK(get(pre + "self_attn.q_proj.weight", y.wqkv, d * d, 0.02f, 0.0f));
        CZCHK(get(pre + "self_attn.k_proj.weight", y.wqkv + d * d, kvd * d, 0.02f, 0.0f));
        CZCHK(get(pre + "self_attn.v_proj.weight", y.wqkv + (d + kvd) * d, kvd * d, 0.02f, 0.0f));
        CZCHK(get(pre + "self_attn.o_proj.weight", y.wo, d * d, 0.02f, 0.0f));
        CZCHK(get(pre + "post_attention_layernorm.weight", y.ln2, d, 0.1f, 1.0f));
        CZCHK(get(pre + "mlp.gate_proj.weight", y.wgu, f * d, 0.02f, 0.0f));
        CZCHK(get(pre + "mlp.up_proj.weight", y.wgu + f * d, f * d, 0.02f, 0.0f));
        CZCHK(get(pre + "mlp.down_proj.weight", y.wd, d * f, 0.02f, 0.0f));
    }
    CZCHK(get("model.norm.weight", m.lnf, d, 0.1f, 1.0f));
    // RoPE tables (candle llama Cache::new): inv_freq_i = 1/theta^(2i/64) in f32, angle = pos * inv_freq (f32 product),
    // cos/sin rounded from f64 libm (the reference's f32 libm is not reproducible across hosts)
    const int mp = c.max_pos;
    std::vector<float> hc((size_t)mp * 32), hs((size_t)mp * 32);
    float inv[32];
    for (int i = 0; i < 32; ++i) {
        float e = (float)(2 * i) / 64.0f;
        float pw = (float)pow((double)c.rope_theta, (double)e);
        inv[i] = 1.0f / pw;
    }
    for (int pz = 0; pz < mp; ++pz)
        for (int i = 0; i < 32; ++i) {
            float a = (float)pz * inv[i];
            hc[(size_t)pz * 32 + i] = (float)cos((double)a);
            hs[(size_t)pz * 32 + i] = (float)sin((double)a);
        }
    HIPCHK(hipMalloc((void**)&m.rope_cos, hc.size() * 4));
    HIPCHK(hipMalloc((void**)&m.rope_sin, hs.size() * 4));
    HIPCHK(hipMemcpy(m.rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    // decode-path copy of the matrices with K in kq order (kernels_decode.hip)
    HIPCHK(hipMalloc((void**)&m.blob_dec, total * 2));
    auto dec = [&](const uint16_t* w) { return m.blob_dec + (w - m.blob); };
    HIPCHK(cz_launch_dec_pack(s->stream, m.E, dec(m.E), V, (int)d));
    for (size_t l = 0; l < L; ++l) {
        const LlamaModel::Layer& y = m.L[l];
        HIPCHK(cz_launch_dec_pack(s->stream, y.wqkv, dec(y.wqkv), d + 2 * kvd, (int)d));
        HIPCHK(cz_launch_dec_pack(s->stream, y.wo, dec(y.wo), d, (int)d));
        HIPCHK(cz_launch_dec_pack(s->stream, y.wgu, dec(y.wgu), 2 * f, (int)d));
        HIPCHK(cz_launch_dec_pack(s->stream, y.wd, dec(y.wd), d, (int)f));
    }
    // big-M path: gate|up rows interleaved so that the GEMM epilogue can apply SwiGLU (kernels_gemm.hip)
    for (size_t l = 0; l < L; ++l) m.L[l].wgu_i = nullptr;
    if ((2 * f) % 128 == 0 && f % 32 == 0) {
        HIPCHK(hipMalloc((void**)&m.blob_gu, L * 2 * f * d * 2));
        for (size_t l = 0; l < L; ++l) {
            m.L[l].wgu_i = m.blob_gu + l * 2 * f * d;
            HIPCHK(cz_launch_gu_interleave(s->stream, m.L[l].wgu, m.L[l].wgu_i, (int)f, (int)d));
        }
    }
    // wide lock-step groups: the layer matrices packed for the mid-M GEMM (kernels_gemm_mid.hip), same offsets as blob minus the embedding
    {
        const size_t off0 = (size_t)(m.L[0].ln1 - m.blob);
        HIPCHK(hipMalloc((void**)&m.blob_k2, (total - off0) * 2));
        auto k2 = [&](const uint16_t* w) { return m.blob_k2 + ((size_t)(w - m.blob) - off0); };
        for (size_t l = 0; l < L; ++l) {
            const LlamaModel::Layer& y = m.L[l];
            HIPCHK(cz_launch_k2_pack(s->stream, y.wqkv, k2(y.wqkv), d + 2 * kvd, (int)d, c.n_heads + c.n_kv_heads));   // (RoPE pairs side by side: cz_launch_gemm_mid_qkv)
            HIPCHK(cz_launch_k2_pack(s->stream, y.wo, k2(y.wo), d, (int)d, 0));
            HIPCHK(cz_launch_k2_pack(s->stream, y.wd, k2(y.wd), d, (int)f, 0));
            if (y.wgu_i) HIPCHK(cz_launch_k2_pack(s->stream, y.wgu_i, k2(y.wgu), 2 * f, (int)d, 0));     // (the interleaved gate|up rows, in wgu's slot)
        }
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return CZ_OK;
}

// ------------------------------------------------------------------ the transformer body over M packed rows
// The rows whose logits are wanted (encode: the rows that predict a coded symbol; re-prime: a stream's last row).  In the LAST layer every
// row still needs its K and V (they are what the wanted rows attend to / what the cache keeps), but only the wanted rows need attention
// output, o-projection and MLP: the layer's tail runs on them alone (encode: half the rows of a segment; re-prime: one row in 511).
struct LastRows {
    const cz_seq_desc* seqd_q;   // per sequence: the wanted QUERY rows (row0 = first wanted row, pos0 = its position; k/v as in seqd)
    int max_rows_q;
    const int* rows;             // device: packed index of every wanted row, in the order the head will ask for them
    int n;
};
// tokens are in s->tok; on return s->x holds the residual stream [M][d] — or, with `last` and fewer wanted rows than rows, s->rc holds
// the residual stream of the wanted rows [last->n][d] (returns that pointer in *compact).
static cz_status llama_body(cz_session* s, int M, const cz_seq_desc* seqd, const cz_rope_desc* roped, int nseq, int max_rows,
                            int max_nk, size_t kv_layer_stride, const int* pos_dev, const LastRows* last = nullptr, const float** compact = nullptr) {
    const LlamaModel& m = s->llama; const cz_llama_config& c = m.cfg;
    const int d = c.d_model, kvd = c.n_kv_heads * 64, f = c.d_ffn, nq = d + 2 * kvd;
    if (compact) *compact = nullptr;
    const bool prune = last && compact && last->n > 0 && last->n < M && !s->no_last_prune;
    CZCHK(s->x.ensure((size_t)M * d * 4)); CZCHK(s->xn.ensure((size_t)M * d * 4));
    if (prune) { CZCHK(s->xc.ensure((size_t)last->n * d * 4)); CZCHK(s->rc.ensure((size_t)last->n * d * 4)); }
    const int big = s->force_big_gemm;
    const bool fused_gu = m.L[0].wgu_i && (M > 64 || big);        // SwiGLU in the gate|up GEMM epilogue: the 2f-wide intermediate is never allocated
    const bool fused_gu_q = prune && m.L[0].wgu_i && (last->n > 64 || big);
    CZCHK(s->qkv.ensure((size_t)M * nq * 4));
    if (!fused_gu) CZCHK(s->gu.ensure((size_t)M * 2 * f * 4));
    else if (prune && !fused_gu_q) CZCHK(s->gu.ensure((size_t)last->n * 2 * f * 4));
    CZCHK(s->h.ensure((size_t)M * f * 4));
    float *x = (float*)s->x.p, *xn = (float*)s->xn.p, *qkv = (float*)s->qkv.p, *gu = (float*)s->gu.p, *h = (float*)s->h.p;
    hipStream_t st = s->stream;
    auto gemm = [&](int rows, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int N, int K) -> hipError_t {
        s->prof_begin(rows <= 64 && !big ? PK_GEMM_SKINNY : (N % 128 ? PK_GEMM_N64 : PK_GEMM));      // 128x64 tiles when N % 128 != 0 (kernels_gemm.hip)
        hipError_t e = cz_launch_gemm(st, A, lda, W, C, ldc, R, ldc, rows, N, K, big);
        s->prof_end(2.0 * rows * (double)N * K, (double)N * K * 2 + (double)rows * (N + K) * 4);
        return e;
    };
    // the layer after attention on `rows` rows: xr = attn . Wo^T + xr; xr = MLP(norm(xr)) + xr   (attn is overwritten with the normed rows)
    auto tail = [&](const LlamaModel::Layer& y, int rows, float* attn, float* xr, bool fused) -> cz_status {
        HIPCHK(gemm(rows, attn, d, y.wo, xr, d, xr, d, d));                  // x = attn_out . Wo^T + x
        s->prof_begin(PK_RMSNORM); HIPCHK(cz_launch_rmsnorm(st, xr, d, y.ln2, attn, d, rows, d, c.rms_eps, nullptr)); s->prof_end(0, (double)rows * d * 8);
        if (fused) {                                        // silu(gate) * up in the GEMM epilogue
            s->prof_begin(PK_GEMM);
            HIPCHK(cz_launch_gemm_swiglu(st, attn, d, y.wgu_i, h, f, rows, f, d));
            s->prof_end(4.0 * rows * (double)f * d, (double)f * d * 4 + (double)rows * (f + d) * 4);
        } else {
            HIPCHK(gemm(rows, attn, d, y.wgu, gu, 2 * f, nullptr, 2 * f, d));
            s->prof_begin(PK_SWIGLU); HIPCHK(cz_launch_swiglu(st, gu, h, rows, f)); s->prof_end(0, (double)rows * f * 12);
        }
        HIPCHK(gemm(rows, h, f, y.wd, xr, d, xr, d, f));                     // x = mlp_out + x
        return CZ_OK;
    };
    s->prof_begin(PK_EMBED);
    HIPCHK(cz_launch_embed(st, (const uint32_t*)s->tok.p, m.E, x, M, d));
    s->prof_end(0, (double)M * d * 6);
    for (int l = 0; l < c.n_layers; ++l) {
        const LlamaModel::Layer& y = m.L[l];
        const bool pr = prune && l == c.n_layers - 1;
        s->prof_begin(PK_RMSNORM); HIPCHK(cz_launch_rmsnorm(st, x, d, y.ln1, xn, d, M, d, c.rms_eps, nullptr)); s->prof_end(0, (double)M * d * 8);
        HIPCHK(gemm(M, xn, d, y.wqkv, qkv, nq, nullptr, nq, d));
        s->prof_begin(PK_ROPE);
        HIPCHK(cz_launch_rope(st, qkv, nq, roped, nseq, max_rows, c.n_heads, c.n_kv_heads, m.rope_cos, m.rope_sin, (size_t)l * kv_layer_stride, pos_dev));
        s->prof_end(0, (double)M * nq * 8);
        s->prof_begin(PK_ATTN);
        HIPCHK(cz_launch_attention(st, qkv, nq, xn, d, pr ? last->seqd_q : seqd, nseq, pr ? last->max_rows_q : max_rows, max_nk, c.n_heads, c.n_kv_heads,
                                   (size_t)l * kv_layer_stride, pos_dev));
        s->prof_end(4.0 * 64 * c.n_heads * (double)(pr ? last->n : M) * max_nk * 0.5, 0);
        if (!pr) { CZCHK(tail(y, M, xn, x, fused_gu)); continue; }
        float *xc = (float*)s->xc.p, *rc = (float*)s->rc.p;
        s->prof_begin(PK_EMBED);
        HIPCHK(cz_launch_gather_rows(st, xn, d, last->rows, xc, d, last->n, d));
        HIPCHK(cz_launch_gather_rows(st, x, d, last->rows, rc, d, last->n, d));
        s->prof_end(0, (double)last->n * d * 16);
        CZCHK(tail(y, last->n, xc, rc, fused_gu_q));
        *compact = rc;
    }
    return CZ_OK;
}

// final norm on selected rows + tied lm_head: logits[r][:] for r in [0, nrows) from residual rows row_index[r]
// (xsrc != nullptr: the nrows rows are contiguous there — llama_body's compact residual — and row_index_dev is not used)
static cz_status llama_head(cz_session* s, const int* row_index_dev, int nrows, float* logits, const float* xsrc = nullptr) {
    const LlamaModel& m = s->llama; const cz_llama_config& c = m.cfg;
    const int d = c.d_model;
    CZCHK(s->xl.ensure((size_t)nrows * d * 4));
    s->prof_begin(PK_RMSNORM);
    HIPCHK(cz_launch_rmsnorm(s->stream, xsrc ? xsrc : (const float*)s->x.p, d, m.lnf, (float*)s->xl.p, d, nrows, d, c.rms_eps, xsrc ? nullptr : row_index_dev));
    s->prof_end(0, (double)nrows * d * 8);
    const int big = s->force_big_gemm;
    s->prof_begin(nrows <= 64 && !big ? PK_GEMM_SKINNY : PK_GEMM);
    HIPCHK(cz_launch_gemm(s->stream, (const float*)s->xl.p, d, m.E, logits, c.vocab, nullptr, 0, nrows, c.vocab, d, big));
    s->prof_end(2.0 * nrows * (double)c.vocab * d, (double)c.vocab * d * 2 + (double)nrows * (c.vocab + d) * 4);
    return CZ_OK;
}

// ------------------------------------------------------------------ lock-step streams (KV cache resident in HBM)
// longest context a stream of this model may reach: the reference fails past max_position_embeddings (candle's RoPE tables end there,
// src/models.rs:94 -> Llama::forward); the attention kernels take CZ_KERNEL_MAX_CTX positions
static inline size_t cz_ctx_limit(const cz_session* s) { return (size_t)std::min(s->llama.cfg.max_pos, CZ_KERNEL_MAX_CTX); }
// old cache [stream][layer][K^T [kvd][oc] | V [oc][kvd]] -> the same layout with nc >= oc positions per stream
__global__ void kv_regrow_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n_sl /* streams x layers */, int kvd, size_t oc, size_t nc) {
    const size_t per = 2 * oc * kvd;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_sl * per; i += (size_t)gridDim.x * blockDim.x) {
        const size_t sl = i / per, r = i % per;
        size_t o;
        if (r < oc * kvd) { const size_t d = r / oc, j = r % oc; o = d * nc + j; }     // K transposed: row d keeps its keys 0..oc-1
        else o = nc * kvd + (r - oc * kvd);                                             // V rows are contiguous
        dst[sl * 2 * nc * kvd + o] = src[i];
    }
}
// the lock-step streams' resident state; need_ctx = positions per stream the caller is about to use (0: whatever is there).  A cache that
// is too short is replaced by a longer one (preserve: with its contents, for a single-stream session that steps past it); captured step
// graphs hold the old strides, so devbuf_gen moves with it.
static cz_status ensure_streams(cz_session* s, int B, size_t need_ctx = 0, bool preserve = false) {
    const cz_llama_config& c = s->llama.cfg;
    const size_t kvd = (size_t)c.n_kv_heads * 64;
    if (B > s->max_streams) { cz_set_error("%d streams exceed max_streams = %d", B, s->max_streams); return CZ_ERR_ARG; }
    if (need_ctx > cz_ctx_limit(s)) {
        cz_set_error("a stream would reach %zu positions: beyond max_position_embeddings = %d (the reference fails there too) / the kernels' %d", need_ctx, c.max_pos, CZ_KERNEL_MAX_CTX);
        return CZ_ERR_STATE;
    }
    const bool fresh = !s->kvcache;
    if (fresh || need_ctx > s->ctx_cap) {
        size_t cap = s->kv_ctx_cap ? (size_t)s->kv_ctx_cap : (size_t)CZ_DEF_CTX;
        if (!fresh && preserve) cap = std::max(cap, 2 * s->ctx_cap);              // a stepping stream grows geometrically
        cap = std::max(cap, (need_ctx + 255) / 256 * 256);
        cap = std::min(cap, (cz_ctx_limit(s) + 255) / 256 * 256);
        const size_t layer_stride = 2 * cap * kvd, stream_stride = (size_t)c.n_layers * layer_stride;
        if (!fresh && !preserve) {
            // nothing of the old cache is kept: give it back BEFORE asking for the longer one (512 streams x 30 layers x 2048 positions are 48 GB
            // per lane; holding both would fail where the new size alone fits)
            HIPCHK(hipStreamSynchronize(s->stream));
            HIPCHK(hipFree(s->kvcache));
            s->kvcache = nullptr; s->ctx_cap = 0;
            ++s->devbuf_gen;                                                     // captured graphs hold kv offsets of the old capacity
        }
        float* nk = nullptr;
        if (hipMalloc((void**)&nk, (size_t)s->max_streams * stream_stride * 4) != hipSuccess) {
            (void)hipGetLastError();
            cz_set_error("cannot allocate the KV cache: %d streams x %zu positions = %.1f GB (lower max_streams, or the re-prime interval)", s->max_streams, cap,
                         (double)s->max_streams * stream_stride * 4 / 1e9);
            return CZ_ERR_HIP;
        }
        if (!fresh && preserve) {                                                // a stepping stream keeps its keys: allocate, copy, free
            HIPCHK(hipStreamSynchronize(s->stream));
            hipLaunchKernelGGL(kv_regrow_kernel, dim3(2048), dim3(256), 0, s->stream, s->kvcache, nk, (size_t)s->max_streams * c.n_layers, (int)kvd, s->ctx_cap, cap);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s->stream));
            HIPCHK(hipFree(s->kvcache));
            ++s->devbuf_gen;
        }
        s->kvcache = nk; s->ctx_cap = cap; s->kv_layer_stride = layer_stride; s->kv_stream_stride = stream_stride;
        std::vector<cz_seq_desc> sd(s->max_streams); std::vector<cz_rope_desc> rd(s->max_streams);
        for (int b = 0; b < s->max_streams; ++b) {
            float* kb = s->kvcache + (size_t)b * s->kv_stream_stride;
            sd[b] = {b, 1, 0, (int)kvd, kb, kb + s->ctx_cap * kvd, (int)s->ctx_cap, 0};     // K transposed [kvd][ctx_cap], V rows [ctx_cap][kvd]
            rd[b] = {b, 1, 0, (int)kvd, kb, kb + s->ctx_cap * kvd, (int)s->ctx_cap, 0};
        }
        const bool first = !s->pos_dev;                                          // (a cache given back above and not replaced leaves these in place)
        if (first) {
            HIPCHK(hipMalloc((void**)&s->pos_dev, 64));
            HIPCHK(hipMalloc((void**)&s->step_seqd, sd.size() * sizeof(cz_seq_desc)));
            HIPCHK(hipMalloc((void**)&s->step_roped, rd.size() * sizeof(cz_rope_desc)));
        }
        // (never the legacy stream here: another decode lane may be capturing its step graph at this moment)
        HIPCHK(hipMemcpyAsync(s->step_seqd, sd.data(), sd.size() * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->step_roped, rd.data(), rd.size() * sizeof(cz_rope_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (first) {
            std::vector<int> ri(s->max_streams); for (int b = 0; b < s->max_streams; ++b) ri[b] = b;
            HIPCHK(hipMalloc((void**)&s->step_rows, ri.size() * 4));
            HIPCHK(hipMemcpyAsync(s->step_rows, ri.data(), ri.size() * 4, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            HIPCHK(hipMalloc((void**)&s->step_logits, (size_t)s->max_streams * c.vocab * 4));
            HIPCHK(hipMalloc((void**)&s->step_sym, (size_t)s->max_streams * 4));
            s->attn_split_streams = std::min(s->max_streams, 32);                 // (groups of more streams than that never take the split form)
            HIPCHK(hipMalloc((void**)&s->attn_split, cz_dec_attn_split_bytes(s->attn_split_streams, c.n_heads)));
        }
    }
    return CZ_OK;
}
// the lock-step decode step on the dedicated latency-oriented kernels (kernels_decode.hip)
static cz_status llama_step_fast(cz_session* s, int B, int pos, int max_nk) {
    const LlamaModel& m = s->llama; const cz_llama_config& c = m.cfg;
    const int d = c.d_model, f = c.d_ffn;
    CZCHK(s->x.ensure((size_t)B * d * 4)); CZCHK(s->xn.ensure((size_t)B * d * 4)); CZCHK(s->qkv.ensure((size_t)B * d * 4));
    CZCHK(s->gu.ensure((size_t)B * d * 4)); CZCHK(s->h.ensure((size_t)B * f * 4));
    float *x = (float*)s->x.p, *xn = (float*)s->xn.p, *q = (float*)s->qkv.p, *att = (float*)s->gu.p, *h = (float*)s->h.p;
    hipStream_t st = s->stream;
    auto dec = [&](const uint16_t* w) { return m.blob_dec + (w - m.blob); };
    s->prof_begin(PK_EMBED); HIPCHK(cz_launch_embed(st, (const uint32_t*)s->tok.p, m.E, x, B, d)); s->prof_end(0, (double)B * d * 6);
    const int kvd = c.n_kv_heads * 64, nq = d + 2 * kvd;
    for (int l = 0; l < c.n_layers; ++l) {
        const LlamaModel::Layer& y = m.L[l];
        const size_t kv_off = (size_t)l * s->kv_layer_stride;
        if (!(s->dbg_skip & 2)) {
        s->prof_begin(PK_DEC_GEMM_QKV);                                                    // RMSNorm(ln1) fused into the A load
        HIPCHK(cz_launch_dec_gemm_qkv(st, x, d, dec(y.wqkv), q, d, B, d, c.n_heads, c.n_kv_heads, s->step_roped, kv_off, s->pos_dev, m.rope_cos, m.rope_sin,
                                      y.ln1, c.rms_eps));
        s->prof_end(2.0 * B * (double)nq * d, (double)nq * d * 2);
        }
        if (!(s->dbg_skip & 1)) {
        s->prof_begin(PK_DEC_ATTN);
        HIPCHK(cz_launch_dec_attn(st, q, d, att, d, s->step_seqd, B, max_nk, c.n_heads, c.n_kv_heads, kv_off, s->pos_dev, 1, s->attn_split, s->attn_split_streams, s->lanes_concurrent));
        s->prof_end(4.0 * 64 * c.n_heads * (double)B * (pos + 1), 2.0 * B * (double)(pos + 1) * kvd * 4);
        }
        if (s->dbg_skip & 2) continue;
        s->prof_begin(PK_DEC_GEMM); HIPCHK(cz_launch_dec_gemm_plain(st, att, d, dec(y.wo), x, d, x, B, d, d, nullptr, 0.0f)); s->prof_end(2.0 * B * (double)d * d, (double)d * d * 2);
        if ((s->fuse_norm || B <= 16) && cz_dec_gemm_norm_is_cheap(d)) {   // RMSNorm(ln2) inside the gate|up projection (sum of squares on the MFMA): one launch less per
                                                                          // layer; pays for one row tile (a single stream: -3 % per step), loses from 128 rows on (+17 us per layer)
            s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_dec_gemm_swiglu(st, x, d, dec(y.wgu), h, f, B, f, d, y.ln2, c.rms_eps)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
        } else {                                                   // many column blocks would each repeat a pass over their rows: normalise once
            s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_dec_rmsnorm(st, x, d, y.ln2, xn, d, B, d, c.rms_eps)); s->prof_end(0, (double)B * d * 8);
            s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_dec_gemm_swiglu(st, xn, d, dec(y.wgu), h, f, B, f, d, nullptr, 0.0f)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
        }
        s->prof_begin(PK_DEC_GEMM); HIPCHK(cz_launch_dec_gemm_plain(st, h, f, dec(y.wd), x, d, x, B, d, f, nullptr, 0.0f)); s->prof_end(2.0 * B * (double)d * f, (double)d * f * 2);
    }
    s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_dec_rmsnorm(st, x, d, m.lnf, xn, d, B, d, c.rms_eps)); s->prof_end(0, (double)B * d * 8);
    s->prof_begin(PK_DEC_GEMM_HEAD);
    HIPCHK(cz_launch_dec_gemm_plain(st, xn, d, dec(m.E), s->step_logits, c.vocab, nullptr, B, c.vocab, d, nullptr, 0.0f));
    s->prof_end(2.0 * B * (double)c.vocab * d, (double)c.vocab * d * 2 + (double)B * c.vocab * 4);
    return CZ_OK;
}
// The same step for a HANDFUL of streams (the CLI's one stream per file): every projection on the VALU kernels of kernels_gemv.hip — one wave per
// block, no block barrier, one memory round trip per launch — with natural-layout activations; attention as in llama_step_fast.  Five launches per
// layer (the norms ride in the QKV / gate|up / lm_head projections).  Same canonical chains, same bits.
static cz_status llama_step_gemv(cz_session* s, int B, int pos, int max_nk) {
    const LlamaModel& m = s->llama; const cz_llama_config& c = m.cfg;
    const int d = c.d_model, f = c.d_ffn;
    CZCHK(s->x.ensure((size_t)B * d * 4)); CZCHK(s->qkv.ensure((size_t)B * d * 4)); CZCHK(s->gu.ensure((size_t)B * d * 4)); CZCHK(s->h.ensure((size_t)B * f * 4));
    float *x = (float*)s->x.p, *q = (float*)s->qkv.p, *att = (float*)s->gu.p, *h = (float*)s->h.p;
    hipStream_t st = s->stream;
    s->prof_begin(PK_EMBED); HIPCHK(cz_launch_embed(st, (const uint32_t*)s->tok.p, m.E, x, B, d)); s->prof_end(0, (double)B * d * 6);
    const int kvd = c.n_kv_heads * 64, nq = d + 2 * kvd;
    for (int l = 0; l < c.n_layers; ++l) {
        const LlamaModel::Layer& y = m.L[l];
        const size_t kv_off = (size_t)l * s->kv_layer_stride;
        if (!(s->dbg_skip & 2)) {
            s->prof_begin(PK_DEC_GEMM_QKV);
            HIPCHK(cz_launch_dec_gemv_qkv(st, x, d, y.wqkv, q, d, B, d, c.n_heads, c.n_kv_heads, s->step_roped, kv_off, s->pos_dev, m.rope_cos, m.rope_sin, y.ln1, c.rms_eps));
            s->prof_end(2.0 * B * (double)nq * d, (double)nq * d * 2);
        }
        // the split (three-launch) attention of a handful of streams leaves the ordered sum of its chunk partials to the o-projection: one launch less per layer
        const bool fold = g_gemv_fold_attn && !s->dbg_skip && d <= 768 && cz_dec_attn_is_split(B, max_nk, c.n_heads, c.n_kv_heads, s->attn_split, s->attn_split_streams);
        if (!(s->dbg_skip & 1)) {
            s->prof_begin(PK_DEC_ATTN);
            HIPCHK(cz_launch_dec_attn(st, q, d, att, d, s->step_seqd, B, max_nk, c.n_heads, c.n_kv_heads, kv_off, s->pos_dev, fold ? -1 : 0, s->attn_split, s->attn_split_streams, s->lanes_concurrent));
            s->prof_end(4.0 * 64 * c.n_heads * (double)B * (pos + 1), 2.0 * B * (double)(pos + 1) * kvd * 4);
        }
        if (s->dbg_skip & 2) continue;
        s->prof_begin(PK_DEC_GEMM);
        if (fold) {
            int maxc = 0;
            const float* part = cz_dec_attn_split_partials(s->attn_split, s->attn_split_streams, c.n_heads, &maxc);
            HIPCHK(cz_launch_dec_gemv_attsum(st, part, maxc, max_nk, s->pos_dev, y.wo, x, d, x, B, d, c.n_heads));
        } else HIPCHK(cz_launch_dec_gemv_plain(st, att, d, y.wo, x, d, x, B, d, d, nullptr, 0.0f));
        s->prof_end(2.0 * B * (double)d * d, (double)d * d * 2);
        s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_dec_gemv_swiglu(st, x, d, y.wgu, h, f, B, f, d, y.ln2, c.rms_eps)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
        s->prof_begin(PK_DEC_GEMM); HIPCHK(cz_launch_dec_gemv_plain(st, h, f, y.wd, x, d, x, B, d, f, nullptr, 0.0f)); s->prof_end(2.0 * B * (double)d * f, (double)d * f * 2);
    }
    s->prof_begin(PK_DEC_GEMM_HEAD);
    HIPCHK(cz_launch_dec_gemv_plain(st, x, d, m.E, s->step_logits, c.vocab, nullptr, B, c.vocab, d, m.lnf, c.rms_eps));
    s->prof_end(2.0 * B * (double)c.vocab * d, (double)c.vocab * d * 2 + (double)B * c.vocab * 4);
    return CZ_OK;
}
// The same step for ONE WIDE lock-step group (a few hundred to 1024 streams): the projections go through the LDS-tiled MFMA GEMM with
// 64-row tiles instead of the latency-oriented dec_gemm_kernel (whose blocks re-read the activations once per 32 output columns and
// reach 10-30 % of the MFMA rate at these row counts); attention is the same dec_attn_kernel (natural-layout output), HBM-bound on the
// KV cache and, with B x n_kv >> 256 blocks, evenly spread over the CUs.  Same canonical arithmetic, same bits.
static cz_status llama_step_wide(cz_session* s, int B, int pos, int max_nk) {
    const LlamaModel& m = s->llama; const cz_llama_config& c = m.cfg;
    const int d = c.d_model, f = c.d_ffn, kvd = c.n_kv_heads * 64, nq = d + 2 * kvd;
    CZCHK(s->x.ensure((size_t)B * d * 4)); CZCHK(s->xn.ensure((size_t)B * d * 4)); CZCHK(s->qkv.ensure((size_t)B * nq * 4)); CZCHK(s->h.ensure((size_t)B * f * 4));
    const bool fused_gu = m.L[0].wgu_i != nullptr;
    const bool hybrid = s->wide_mode == 2;
    if (!fused_gu && !hybrid) CZCHK(s->gu.ensure((size_t)B * 2 * f * 4));
    float *x = (float*)s->x.p, *xn = (float*)s->xn.p, *qkv = (float*)s->qkv.p, *gu = (float*)s->gu.p, *h = (float*)s->h.p;
    hipStream_t st = s->stream;
    const bool mid = s->wide_mode == 3 && fused_gu && m.blob_k2;     // the K-group GEMM (kernels_gemm_mid.hip) for the four layer projections
    const size_t off0 = (size_t)(m.L[0].ln1 - m.blob);
    auto k2 = [&](const uint16_t* w) { return m.blob_k2 + ((size_t)(w - m.blob) - off0); };
    auto gemm_mid = [&](int kind, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int N, int K) -> hipError_t {
        s->prof_begin(kind);
        hipError_t e = cz_launch_gemm_mid(st, A, lda, k2(W), C, ldc, R, ldc, B, N, K);
        s->prof_end(2.0 * B * (double)N * K, (double)N * K * 2 + (double)B * (N + K) * 4);
        return e;
    };
    auto dec = [&](const uint16_t* w) { return m.blob_dec + (w - m.blob); };
    auto gemm = [&](int kind, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int N, int K) -> hipError_t {
        s->prof_begin(kind);
        hipError_t e = cz_launch_gemm(st, A, lda, W, C, ldc, R, ldc, B, N, K, 1);
        s->prof_end(2.0 * B * (double)N * K, (double)N * K * 2 + (double)B * (N + K) * 4);
        return e;
    };
    s->prof_begin(PK_EMBED); HIPCHK(cz_launch_embed(st, (const uint32_t*)s->tok.p, m.E, x, B, d)); s->prof_end(0, (double)B * d * 6);
    for (int l = 0; l < c.n_layers; ++l) {
        const LlamaModel::Layer& y = m.L[l];
        const size_t kv_off = (size_t)l * s->kv_layer_stride;
        if (!(s->dbg_skip & 2)) {
            s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_rmsnorm(st, x, d, y.ln1, xn, d, B, d, c.rms_eps, nullptr)); s->prof_end(0, (double)B * d * 8);
            if (mid) {                                                                     // RoPE and the cache write ride in the projection's epilogue
                s->prof_begin(PK_DEC_GEMM_QKV);
                HIPCHK(cz_launch_gemm_mid_qkv(st, xn, d, k2(y.wqkv), qkv, nq, B, d, c.n_heads, c.n_kv_heads, s->step_roped, kv_off, s->pos_dev, m.rope_cos, m.rope_sin));
                s->prof_end(2.0 * B * (double)nq * d, (double)nq * d * 2 + (double)B * (nq + d) * 4);
            } else {
                HIPCHK(gemm(PK_DEC_GEMM_QKV, xn, d, y.wqkv, qkv, nq, nullptr, nq, d));
                s->prof_begin(PK_ROPE);                                                    // q rotated in place; k (rotated) and v go to the cache rows at *pos_dev
                HIPCHK(cz_launch_rope(st, qkv, nq, s->step_roped, B, 1, c.n_heads, c.n_kv_heads, m.rope_cos, m.rope_sin, kv_off, s->pos_dev));
                s->prof_end(0, (double)B * nq * 8);
            }
        }
        if (!(s->dbg_skip & 1)) {
            s->prof_begin(PK_DEC_ATTN);
            HIPCHK(cz_launch_dec_attn(st, qkv, nq, xn, d, s->step_seqd, B, max_nk, c.n_heads, c.n_kv_heads, kv_off, s->pos_dev, hybrid ? 1 : 0, s->attn_split, s->attn_split_streams, s->lanes_concurrent));
            s->prof_end(4.0 * 64 * c.n_heads * (double)B * (pos + 1), 2.0 * B * (double)(pos + 1) * kvd * 4);
        }
        if (s->dbg_skip & 2) continue;
        if (hybrid) {     // o-projection, gate|up and down-projection on dec_gemm_kernel's 32 x 64 blocks (K split over the waves of a block): at a few
                          // hundred rows they beat the tiled kernel, whose 64 x 64 tiles walk K serially (the down-projection has 24 K tiles)
            s->prof_begin(PK_DEC_GEMM); HIPCHK(cz_launch_dec_gemm_plain(st, xn, d, dec(y.wo), x, d, x, B, d, d, nullptr, 0.0f)); s->prof_end(2.0 * B * (double)d * d, (double)d * d * 2);
            s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_dec_rmsnorm(st, x, d, y.ln2, xn, d, B, d, c.rms_eps)); s->prof_end(0, (double)B * d * 8);
            s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_dec_gemm_swiglu(st, xn, d, dec(y.wgu), h, f, B, f, d, nullptr, 0.0f)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
            s->prof_begin(PK_DEC_GEMM); HIPCHK(cz_launch_dec_gemm_plain(st, h, f, dec(y.wd), x, d, x, B, d, f, nullptr, 0.0f)); s->prof_end(2.0 * B * (double)d * f, (double)d * f * 2);
            continue;
        }
        if (mid) {
            HIPCHK(gemm_mid(PK_DEC_GEMM, xn, d, y.wo, x, d, x, d, d));                      // x = attn_out . Wo^T + x
            s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_rmsnorm(st, x, d, y.ln2, xn, d, B, d, c.rms_eps, nullptr)); s->prof_end(0, (double)B * d * 8);
            s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_gemm_mid_swiglu(st, xn, d, k2(y.wgu), h, f, B, f, d)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
            HIPCHK(gemm_mid(PK_DEC_GEMM, h, f, y.wd, x, d, x, d, f));                       // x = mlp_out + x
            continue;
        }
        HIPCHK(gemm(PK_DEC_GEMM, xn, d, y.wo, x, d, x, d, d));                              // x = attn_out . Wo^T + x
        s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_rmsnorm(st, x, d, y.ln2, xn, d, B, d, c.rms_eps, nullptr)); s->prof_end(0, (double)B * d * 8);
        if (fused_gu) {
            s->prof_begin(PK_DEC_GEMM_GU); HIPCHK(cz_launch_gemm_swiglu(st, xn, d, y.wgu_i, h, f, B, f, d)); s->prof_end(4.0 * B * (double)f * d, (double)f * d * 4);
        } else {
            HIPCHK(gemm(PK_DEC_GEMM_GU, xn, d, y.wgu, gu, 2 * f, nullptr, 2 * f, d));
            s->prof_begin(PK_SWIGLU); HIPCHK(cz_launch_swiglu(st, gu, h, B, f)); s->prof_end(0, (double)B * f * 12);
        }
        HIPCHK(gemm(PK_DEC_GEMM, h, f, y.wd, x, d, x, d, f));                               // x = mlp_out + x
    }
    s->prof_begin(PK_DEC_NORM); HIPCHK(cz_launch_rmsnorm(st, x, d, m.lnf, xn, d, B, d, c.rms_eps, nullptr)); s->prof_end(0, (double)B * d * 8);
    HIPCHK(gemm(PK_DEC_GEMM_HEAD, xn, d, m.E, s->step_logits, c.vocab, nullptr, c.vocab, d));
    return CZ_OK;
}
// the lock-step decode step: the latency-oriented kernels for narrow groups, the tiled-GEMM chain for wide ones
static cz_status llama_step_any(cz_session* s, int B, int pos, int max_nk) {
    if (s->wide_min_streams > 0 && B >= s->wide_min_streams) return llama_step_wide(s, B, pos, max_nk);
    if (cz_dec_gemv_ok(B, s->llama.cfg.d_model, true) && cz_dec_gemv_ok(B, s->llama.cfg.d_ffn, false)) return llama_step_gemv(s, B, pos, max_nk);
    return llama_step_fast(s, B, pos, max_nk);
}
// one token per stream at position `pos` (tokens in s->tok[0..B)); logits -> s->step_logits [B][V]
static cz_status llama_step(cz_session* s, int B, int pos) {
    if (pos >= (int)cz_ctx_limit(s) || (s->ctx_cap && pos >= (int)s->ctx_cap)) { cz_set_error("position %d exceeds the context (max_position_embeddings %d, KV cache %zu)", pos, s->llama.cfg.max_pos, s->ctx_cap); return CZ_ERR_STATE; }
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev, pos);
    if (!s->decode_v1 && s->llama.cfg.n_heads / s->llama.cfg.n_kv_heads <= 4) return llama_step_any(s, B, pos, pos + 1);
    if (B > 64) s->force_big_gemm = 1;                     // the general skinny kernel covers M <= 64 only
    CZCHK(llama_body(s, B, s->step_seqd, s->step_roped, B, 1, pos + 1, s->kv_layer_stride, s->pos_dev));
    return llama_head(s, s->step_rows, B, s->step_logits);
}
// re-prime: every stream forwards hl tokens (s->tok[b*hl + j]) at positions 0..hl-1 into its cache; logits of the last row
static cz_status llama_reprime(cz_session* s, int B, int hl) {
    if (hl <= 0 || hl > (int)cz_ctx_limit(s) || (s->ctx_cap && hl > (int)s->ctx_cap)) { cz_set_error("history length %d out of range", hl); return CZ_ERR_ARG; }
    const size_t kvd = (size_t)s->llama.cfg.n_kv_heads * 64;
    std::vector<cz_seq_desc> sd(B), sq(B); std::vector<cz_rope_desc> rd(B); std::vector<int> ri(B);
    for (int b = 0; b < B; ++b) {
        float* kb = s->kvcache + (size_t)b * s->kv_stream_stride;
        sd[b] = {b * hl, hl, 0, (int)kvd, kb, kb + s->ctx_cap * kvd, (int)s->ctx_cap, 0};
        sq[b] = {b * hl + hl - 1, 1, hl - 1, (int)kvd, kb, kb + s->ctx_cap * kvd, (int)s->ctx_cap, 0};     // the last row: the only one whose logits are used
        rd[b] = {b * hl, hl, 0, (int)kvd, kb, kb + s->ctx_cap * kvd, (int)s->ctx_cap, 0};
        ri[b] = b * hl + hl - 1;
    }
    CZCHK(s->seqd.ensure(sd.size() * sizeof(cz_seq_desc))); CZCHK(s->roped.ensure(rd.size() * sizeof(cz_rope_desc)));
    CZCHK(s->rowidx.ensure(ri.size() * 4)); CZCHK(s->seqd_q.ensure(sq.size() * sizeof(cz_seq_desc)));
    HIPCHK(hipMemcpyAsync(s->seqd_q.p, sq.data(), sq.size() * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->seqd.p, sd.data(), sd.size() * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->roped.p, rd.data(), rd.size() * sizeof(cz_rope_desc), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->rowidx.p, ri.data(), ri.size() * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));       // host vectors go out of scope
    const LastRows last{(const cz_seq_desc*)s->seqd_q.p, 1, (const int*)s->rowidx.p, B};
    const float* xq = nullptr;
    CZCHK(llama_body(s, B * hl, (const cz_seq_desc*)s->seqd.p, (const cz_rope_desc*)s->roped.p, B, hl, hl, s->kv_layer_stride, nullptr, &last, &xq));
    return llama_head(s, (const int*)s->rowidx.p, B, s->step_logits, xq);
}

// size every packed-row buffer for M rows (and nlog logit rows) now, so that nothing moves later
static cz_status reserve_rows(cz_session* s, size_t M, size_t nlog) {
    const cz_llama_config& c = s->llama.cfg;
    const size_t d = c.d_model, kvd = (size_t)c.n_kv_heads * 64, f = c.d_ffn, nq = d + 2 * kvd;
    CZCHK(s->x.ensure(M * d * 4)); CZCHK(s->xn.ensure(M * d * 4)); CZCHK(s->qkv.ensure(M * nq * 4));
    if (!(s->llama.L[0].wgu_i && (M > 64 || s->force_big_gemm))) CZCHK(s->gu.ensure(M * 2 * f * 4));
    CZCHK(s->h.ensure(M * f * 4)); CZCHK(s->xl.ensure(nlog * d * 4));
    CZCHK(s->seqd.ensure(nlog * sizeof(cz_seq_desc))); CZCHK(s->roped.ensure(nlog * sizeof(cz_rope_desc))); CZCHK(s->rowidx.ensure(nlog * 4));
    CZCHK(s->seqd_q.ensure(nlog * sizeof(cz_seq_desc))); CZCHK(s->xc.ensure(nlog * d * 4)); CZCHK(s->rc.ensure(nlog * d * 4));   // the last layer's tail on the wanted rows
    if (nlog <= 64 && !s->force_big_gemm) CZCHK(s->gu.ensure(nlog * 2 * f * 4));
    return CZ_OK;
}

// One decode iteration as a hipGraph: decode_symbol(i) -> tok = sym -> step(pos) -> ++i, ++pos.  All per-iteration scalars
// live in HBM (pos_dev), so the same executable graph is replayed for every token; the host only enqueues.
static void destroy_step_graph(cz_session* s) {
    if (s->graph_exec) { hipGraphExecDestroy(s->graph_exec); s->graph_exec = nullptr; }
    if (s->graph) { hipGraphDestroy(s->graph); s->graph = nullptr; }
    s->graph_B = 0;
}
static cz_status build_step_graph(cz_session* s, int B, int max_nk) {
    destroy_step_graph(s);
    s->graph_gen = s->devbuf_gen;
    const cz_llama_config& c = s->llama.cfg;
    cz_dec_state* dst = (cz_dec_state*)s->decst.p;
    const unsigned long long* step_dev = reinterpret_cast<const unsigned long long*>(s->pos_dev + 2);
    const bool prof = s->prof_on; s->prof_on = false;            // no events inside a captured region
    // every buffer the captured kernels touch must exist before capture starts
    cz_status rc = llama_step_any(s, B, 0, max_nk);
    if (rc != CZ_OK) { s->prof_on = prof; return rc; }
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = cz_launch_decode_symbol(s->stream, s->step_logits, c.vocab, dst, s->step_sym, B, c.vocab, 0, 0ull, s->fallback_dev, step_dev, s->cdf_scratch);
    if (e == hipSuccess) e = hipMemcpyAsync(s->tok.p, s->step_sym, (size_t)B * 4, hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) rc = llama_step_any(s, B, 0, max_nk);
    if (e == hipSuccess && rc == CZ_OK) { hipLaunchKernelGGL(inc_counters_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev); e = hipGetLastError(); }
    hipGraph_t g = nullptr;
    hipError_t e2 = hipStreamEndCapture(s->stream, &g);
    s->prof_on = prof;
    if (e != hipSuccess || e2 != hipSuccess || rc != CZ_OK || !g) {
        if (g) hipGraphDestroy(g);
        cz_set_error("hipGraph capture of the decode step failed (%s / %s)", hipGetErrorString(e), hipGetErrorString(e2));
        return CZ_ERR_HIP;
    }
    s->graph = g;
    HIPCHK(hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
    s->graph_B = B; s->graph_max_nk = max_nk;
    return CZ_OK;
}

// ------------------------------------------------------------------ schedule (the reference's re-prime predicate)
struct Seg { int stream; size_t hist_start, start_i, end_i; };   // rows = ids[hist_start .. end_i), coded i in [start_i, end_i)
static void plan_segments(int stream, size_t n_ids, uint32_t context, uint32_t R, std::vector<Seg>* out) {
    if (n_ids < 2) return;
    const size_t ec = context < 511 ? context : 511;            // src/main.rs:1935
    size_t index_pos = 1, seg_start = 0, hist_start = 0;        // after step(BOS) (src/main.rs:1916)
    for (size_t i = 0; i + 1 < n_ids; ++i) {
        if (index_pos >= ec && R != 0 && (i % R) == 0 && i > 0) {   // src/main.rs:2280
            out->push_back({stream, hist_start, seg_start, i});
            const size_t end = 1 + i;
            hist_start = end > ec ? end - ec : 0;               // :2281-2283
            seg_start = i; index_pos = end - hist_start;
        }
        index_pos += 1;                                          // step(sym) (:2345)
    }
    out->push_back({stream, hist_start, seg_start, n_ids - 1});
}

// ------------------------------------------------------------------ encode
// i_begin / i_end (optional, per stream): only the coded tokens i in [i_begin, i_end) are evaluated and out_lo/out_hi are indexed by
// i - i_begin.  A re-prime segment depends on token ids alone, never on another segment's activations, so any range of any
// stream can be computed on any GPU: this is what shards ONE file's encode over ranks.
static cz_status encode_llama(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n, uint32_t context,
                              uint32_t R, const size_t* i_begin, const size_t* i_end, uint32_t* const* out_lo, uint32_t* const* out_hi) {
    const cz_llama_config& c = s->llama.cfg;
    if (R == 0) { cz_set_error("reprime_interval must be > 0"); return CZ_ERR_ARG; }       // the reference would panic on i % 0
    if (context == 0) { cz_set_error("reprime called with empty history (context must be > 0)"); return CZ_ERR_ARG; }   // src/models.rs:108
    std::vector<Seg> segs;
    for (size_t st = 0; st < n_streams; ++st) {
        if (n[st] == 0) { cz_set_error("stream %zu is empty (ids must start with BOS)", st); return CZ_ERR_ARG; }
        for (size_t i = 0; i < n[st]; ++i) if (ids[st][i] >= (uint32_t)c.vocab) { cz_set_error("token id %u >= vocab %d", ids[st][i], c.vocab); return CZ_ERR_ARG; }
        const size_t rb = i_begin ? i_begin[st] : 0, re = i_end ? i_end[st] : n[st] - 1;
        if (rb > re || re > n[st] - 1) { cz_set_error("stream %zu: token range [%zu, %zu) outside [0, %zu)", st, rb, re, n[st] - 1); return CZ_ERR_ARG; }
        std::vector<Seg> all;
        plan_segments((int)st, n[st], context, R, &all);
        for (Seg sg : all) {                                     // clip to the range: rows after the last wanted logits row are not needed (causal)
            if (sg.end_i <= rb || sg.start_i >= re) continue;
            if (sg.start_i < rb) sg.start_i = rb;
            if (sg.end_i > re) sg.end_i = re;
            segs.push_back(sg);
        }
    }
    for (auto& sg : segs) {
        size_t T = sg.end_i - sg.hist_start;
        if (T > cz_ctx_limit(s)) { cz_set_error("a segment of %zu positions exceeds max_position_embeddings = %d (the reference fails there too) / the kernels' %d", T, c.max_pos, CZ_KERNEL_MAX_CTX); return CZ_ERR_STATE; }
    }
    const int d = c.d_model, kvd = c.n_kv_heads * 64, nq = d + 2 * kvd;
    size_t si = 0;
    while (si < segs.size()) {
        // pack segments into one batch
        size_t M = 0, sj = si; int max_rows = 0;
        while (sj < segs.size()) {
            size_t T = segs[sj].end_i - segs[sj].hist_start;
            if (M && M + T > s->max_batch_rows) break;
            M += T; max_rows = std::max(max_rows, (int)T); ++sj;
        }
        const int nseq = (int)(sj - si);
        std::vector<uint32_t> tok(M), sym; std::vector<int> rowidx; std::vector<cz_seq_desc> sd(nseq), sq(nseq); std::vector<cz_rope_desc> rd(nseq);
        int max_rows_q = 0;
        struct Dst { int stream; size_t i; }; std::vector<Dst> dst;
        CZCHK(s->qkv.ensure(M * (size_t)nq * 4));
        float* qkv = (float*)s->qkv.p;
        size_t row = 0;
        for (int q = 0; q < nseq; ++q) {
            const Seg& sg = segs[si + q];
            const size_t T = sg.end_i - sg.hist_start;
            const uint32_t* src = ids[sg.stream];
            for (size_t p = 0; p < T; ++p) tok[row + p] = src[sg.hist_start + p];
            float* kb = qkv + row * nq + d;                     // K/V stay in place inside the fused QKV rows
            sd[q] = {(int)row, (int)T, 0, nq, kb, kb + kvd, 0, 0};
            rd[q] = {(int)row, (int)T, 0, nq, kb, kb + kvd, 0, 0};
            const int q0 = (int)(sg.start_i - sg.hist_start), nqr = (int)(sg.end_i - sg.start_i);     // the rows that predict a coded symbol
            sq[q] = {(int)row + q0, nqr, q0, nq, kb, kb + kvd, 0, 0};
            max_rows_q = std::max(max_rows_q, nqr);
            for (size_t i = sg.start_i; i < sg.end_i; ++i) {
                rowidx.push_back((int)(row + (i - sg.hist_start)));
                sym.push_back(src[i + 1]);
                dst.push_back({sg.stream, i});
            }
            row += T;
        }
        const size_t Lr = rowidx.size();
        CZCHK(s->tok.ensure(M * 4)); CZCHK(s->seqd.ensure(nseq * sizeof(cz_seq_desc))); CZCHK(s->roped.ensure(nseq * sizeof(cz_rope_desc)));
        CZCHK(s->rowidx.ensure(Lr * 4)); CZCHK(s->sym.ensure(Lr * 4)); CZCHK(s->clo.ensure(Lr * 4)); CZCHK(s->chi.ensure(Lr * 4));
        CZCHK(s->seqd_q.ensure(nseq * sizeof(cz_seq_desc)));
        HIPCHK(hipMemcpyAsync(s->tok.p, tok.data(), M * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->seqd.p, sd.data(), nseq * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->seqd_q.p, sq.data(), nseq * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->roped.p, rd.data(), nseq * sizeof(cz_rope_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->rowidx.p, rowidx.data(), Lr * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->sym.p, sym.data(), Lr * 4, hipMemcpyHostToDevice, s->stream));
        const LastRows last{(const cz_seq_desc*)s->seqd_q.p, max_rows_q, (const int*)s->rowidx.p, (int)Lr};
        const float* xq = nullptr;                               // != nullptr: the residual rows of rowidx, contiguous
        CZCHK(llama_body(s, (int)M, (const cz_seq_desc*)s->seqd.p, (const cz_rope_desc*)s->roped.p, nseq, max_rows, max_rows, 0, nullptr, &last, &xq));
        const size_t LB = s->logit_batch_rows;
        CZCHK(s->logits.ensure(std::min(LB, Lr) * (size_t)c.vocab * 4));
        for (size_t r0 = 0; r0 < Lr; r0 += LB) {
            const int nr = (int)std::min(LB, Lr - r0);
            CZCHK(llama_head(s, (const int*)s->rowidx.p + r0, nr, (float*)s->logits.p, xq ? xq + r0 * (size_t)d : nullptr));
            s->prof_begin(PK_CDF);
            if (s->serial_cdf) HIPCHK(cz_launch_cdf_pairs_serial(s->stream, (const float*)s->logits.p, c.vocab, (const uint32_t*)s->sym.p + r0,
                                                                  (uint32_t*)s->clo.p + r0, (uint32_t*)s->chi.p + r0, nr, c.vocab, 0));
            else HIPCHK(cz_launch_cdf_pairs(s->stream, (const float*)s->logits.p, c.vocab, (const uint32_t*)s->sym.p + r0, (uint32_t*)s->clo.p + r0,
                                            (uint32_t*)s->chi.p + r0, nr, c.vocab, 0, s->fallback_dev, s->cdf_scratch));
            s->prof_end(0, (double)nr * c.vocab * 4);
        }
        std::vector<uint32_t> lo(Lr), hi(Lr);
        HIPCHK(hipMemcpyAsync(lo.data(), s->clo.p, Lr * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(hi.data(), s->chi.p, Lr * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        for (size_t r = 0; r < Lr; ++r) {
            const size_t o = dst[r].i - (i_begin ? i_begin[dst[r].stream] : 0);
            out_lo[dst[r].stream][o] = lo[r]; out_hi[dst[r].stream][o] = hi[r];
        }
        si = sj;
    }
    return CZ_OK;
}

// ------------------------------------------------------------------ f4: cross-entropy of target spans (src/main.rs:1725-1751)
// Teacher-forced: every span is ONE causal forward over prime ++ targets[..n-1]; the logits rows that predict the targets go through
// the floor-softmax kernel; -log2 and the running sum are host libm, in the reference's order.  All spans share the GEMMs.
static cz_status xe_llama(cz_session* s, size_t n_spans, const uint32_t* const* hist, const size_t* n_hist, const uint32_t* const* hint, const size_t* n_hint,
                          const uint32_t* const* tgt, const size_t* n_tgt, double* bits_out) {
    const cz_llama_config& c = s->llama.cfg;
    const size_t max_ctx = 512 - 1;                                  // max_context_length().saturating_sub(1)
    struct Span { std::vector<uint32_t> toks; size_t np, nt; };
    std::vector<Span> sp(n_spans);
    for (size_t i = 0; i < n_spans; ++i) {
        bits_out[i] = 0.0;
        const size_t nt = n_tgt[i], nh = n_hist ? n_hist[i] : 0, ng = n_hint ? n_hint[i] : 0;
        sp[i].nt = nt; sp[i].np = 0;
        if (nt == 0) continue;                                        // :1726
        const size_t hb = std::min(ng, max_ctx), ht = std::min(max_ctx - hb, nh);
        if (ht + hb == 0) { cz_set_error("reprime called with empty history"); return CZ_ERR_ARG; }
        auto& v = sp[i].toks;
        v.insert(v.end(), hist[i] + (nh - ht), hist[i] + nh);
        if (hb) v.insert(v.end(), hint[i], hint[i] + hb);
        sp[i].np = v.size();
        v.insert(v.end(), tgt[i], tgt[i] + nt - 1);                   // the last target is scored, never fed
        for (uint32_t t : v) if (t >= (uint32_t)c.vocab) { cz_set_error("token id %u >= vocab %d", t, c.vocab); return CZ_ERR_ARG; }
        if (v.size() > cz_ctx_limit(s)) { cz_set_error("span of %zu positions exceeds the supported context (%zu)", v.size(), cz_ctx_limit(s)); return CZ_ERR_STATE; }
    }
    const int d = c.d_model, kvd = c.n_kv_heads * 64, nq = d + 2 * kvd;
    size_t si = 0;
    while (si < n_spans) {
        size_t M = 0, sj = si; int max_rows = 0;
        while (sj < n_spans) {
            const size_t T = sp[sj].toks.size();
            if (M && M + T > s->max_batch_rows) break;
            M += T; max_rows = std::max(max_rows, (int)T); ++sj;
        }
        if (M == 0) { si = sj; continue; }
        std::vector<uint32_t> tok(M), sym; std::vector<int> rowidx; std::vector<cz_seq_desc> sd; std::vector<cz_rope_desc> rd;
        CZCHK(s->qkv.ensure(M * (size_t)nq * 4));
        float* qkv = (float*)s->qkv.p;
        size_t row = 0;
        for (size_t q = si; q < sj; ++q) {
            const size_t T = sp[q].toks.size();
            if (!T) continue;
            memcpy(tok.data() + row, sp[q].toks.data(), T * 4);
            float* kb = qkv + row * nq + d;
            sd.push_back({(int)row, (int)T, 0, nq, kb, kb + kvd, 0, 0});
            rd.push_back({(int)row, (int)T, 0, nq, kb, kb + kvd, 0, 0});
            for (size_t i = 0; i < sp[q].nt; ++i) { rowidx.push_back((int)(row + sp[q].np - 1 + i)); sym.push_back(tgt[q][i]); }
            row += T;
        }
        const int nseq = (int)sd.size(); const size_t Lr = rowidx.size();
        CZCHK(s->tok.ensure(M * 4)); CZCHK(s->seqd.ensure(nseq * sizeof(cz_seq_desc))); CZCHK(s->roped.ensure(nseq * sizeof(cz_rope_desc)));
        CZCHK(s->rowidx.ensure(Lr * 4)); CZCHK(s->sym.ensure(Lr * 4)); CZCHK(s->clo.ensure(Lr * 8 + 16));     // clo holds the f64 probabilities here
        HIPCHK(hipMemcpyAsync(s->tok.p, tok.data(), M * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->seqd.p, sd.data(), nseq * sizeof(cz_seq_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->roped.p, rd.data(), nseq * sizeof(cz_rope_desc), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->rowidx.p, rowidx.data(), Lr * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->sym.p, sym.data(), Lr * 4, hipMemcpyHostToDevice, s->stream));
        CZCHK(llama_body(s, (int)M, (const cz_seq_desc*)s->seqd.p, (const cz_rope_desc*)s->roped.p, nseq, max_rows, max_rows, 0, nullptr));
        const size_t LB = s->logit_batch_rows;
        CZCHK(s->logits.ensure(std::min(LB, Lr) * (size_t)c.vocab * 4));
        for (size_t r0 = 0; r0 < Lr; r0 += LB) {
            const int nr = (int)std::min(LB, Lr - r0);
            CZCHK(llama_head(s, (const int*)s->rowidx.p + r0, nr, (float*)s->logits.p));
            s->prof_begin(PK_CDF);
            HIPCHK(cz_launch_xe_probs(s->stream, (const float*)s->logits.p, c.vocab, (const uint32_t*)s->sym.p + r0, (double*)s->clo.p + r0, nr, c.vocab, s->fallback_dev,
                                      s->cdf_scratch));
            s->prof_end(0, (double)nr * c.vocab * 4);
        }
        std::vector<double> p(Lr);
        HIPCHK(hipMemcpyAsync(p.data(), s->clo.p, Lr * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        size_t k = 0;
        for (size_t q = si; q < sj; ++q) {
            double bits = 0.0;
            for (size_t i = 0; i < sp[q].nt; ++i) { double pi = p[k++]; if (!(pi > 1e-300)) pi = 1e-300; bits += -log2(pi); }   // bits += -p.max(1e-300).log2()
            bits_out[q] = bits;
        }
        si = sj;
    }
    return CZ_OK;
}

// ------------------------------------------------------------------ decode (lock-step)
// One lock-step group lives on one lane (session): its KV cache, arithmetic-decoder states, decoded ids and position counters stay
// in HBM between calls, so the loop of decode_file (src/main.rs:2528-2541, 2621-2628) can be run in slices: begin -> advance(n)* -> fetch.
// shared by both back-ends: payload buffer + decoder states of a lock-step group.  caps == nullptr: the payloads are complete; else
// stream b's buffer has room for caps[b] bytes and more arrive through dec_run_feed
cz_status cz_dec_run_setup(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* tcs) {
    DecRun& run = s->run;
    size_t pay_total = 0, ids_total = 0, max_tc = 0;
    run.pay_off.assign(B, 0); run.pay_len.assign(lens, lens + B); run.pay_cap.assign(B, 0); run.partial = caps != nullptr;
    for (int b = 0; b < B; ++b) {
        const size_t cap = caps ? std::max(caps[b], lens[b]) : lens[b];
        run.pay_off[b] = pay_total; run.pay_cap[b] = cap;
        pay_total += cap + 16; ids_total += tcs[b] + 1; max_tc = std::max(max_tc, tcs[b]);
    }
    CZCHK(s->payload.ensure(pay_total)); CZCHK(s->outids.ensure(ids_total * 4)); CZCHK(s->decst.ensure(B * sizeof(cz_dec_state)));
    CZCHK(s->sym.ensure((size_t)B * 8));                                  // (new lengths of a feed)
    std::vector<cz_dec_state> hs(B); std::vector<uint8_t> pay(pay_total, 0);
    size_t io = 0;
    for (int b = 0; b < B; ++b) {
        if (lens[b]) memcpy(pay.data() + run.pay_off[b], payloads[b], lens[b]);
        cz_dec_state d; memset(&d, 0, sizeof d);
        d.in = (const uint8_t*)s->payload.p + run.pay_off[b]; d.in_len = lens[b]; d.out_ids = (uint32_t*)s->outids.p + io; d.token_count = tcs[b];
        d.active = tcs[b] > 0; d.final_input = caps ? 0 : 1;
        hs[b] = d; io += tcs[b] + 1;
    }
    HIPCHK(hipMemcpyAsync(s->payload.p, pay.data(), pay_total, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(s->decst.p, hs.data(), B * sizeof(cz_dec_state), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    run.B = B; run.tcs.assign(tcs, tcs + B); run.max_tc = max_tc; run.ids_total = ids_total;
    return CZ_OK;
}
// append bytes to the payloads of this lane's group (between two advances)
cz_status cz_dec_run_feed(cz_session* s, const uint8_t* const* chunks, const size_t* lens, int final_input) {
    DecRun& run = s->run;
    if (!run.active) { cz_set_error("no decode in progress on this session (cz_decode_begin_batch first)"); return CZ_ERR_STATE; }
    if (!run.partial) { cz_set_error("cz_decode_feed_batch: this decode was begun with complete payloads"); return CZ_ERR_STATE; }
    std::vector<unsigned long long> nl(run.B);
    for (int b = 0; b < run.B; ++b) {
        if (run.pay_len[b] + lens[b] > run.pay_cap[b]) { cz_set_error("cz_decode_feed_batch: stream %d would hold %zu bytes, capacity %zu", b, run.pay_len[b] + lens[b], run.pay_cap[b]); return CZ_ERR_ARG; }
        if (lens[b]) HIPCHK(hipMemcpyAsync((uint8_t*)s->payload.p + run.pay_off[b] + run.pay_len[b], chunks[b], lens[b], hipMemcpyHostToDevice, s->stream));
        run.pay_len[b] += lens[b]; nl[b] = run.pay_len[b];
    }
    HIPCHK(hipMemcpyAsync(s->sym.p, nl.data(), run.B * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(cz_launch_dec_feed(s->stream, (cz_dec_state*)s->decst.p, run.B, (const unsigned long long*)s->sym.p, final_input));
    HIPCHK(hipStreamSynchronize(s->stream));                              // the host chunks may go away
    return CZ_OK;
}
// ids decoded so far + the underrun check of a fed decode
cz_status cz_dec_run_fetch(cz_session* s, uint32_t* const* out_ids) {
    DecRun& run = s->run;
    if (!run.active) { cz_set_error("no decode in progress on this session (cz_decode_begin_batch first)"); return CZ_ERR_STATE; }
    std::vector<uint32_t> all(run.ids_total); std::vector<cz_dec_state> hs(run.B);
    HIPCHK(hipMemcpyAsync(all.data(), s->outids.p, run.ids_total * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(hs.data(), s->decst.p, run.B * sizeof(cz_dec_state), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    size_t io = 0;
    for (int b = 0; b < run.B; ++b) { memcpy(out_ids[b], all.data() + io, (run.tcs[b] + 1) * 4); if (run.tcs[b] == 0) out_ids[b][0] = run.bos; io += run.tcs[b] + 1; }
    for (int b = 0; b < run.B; ++b)
        if (hs[b].underrun) { cz_set_error("stream %d: the decoder read past the payload bytes fed so far (advance only as far as the bytes that have arrived allow)", b); return CZ_ERR_STATE; }
    return CZ_OK;
}
static cz_status decode_llama_begin(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* tcs,
                                    uint32_t context_window, uint32_t R, uint32_t bos) {
    const cz_llama_config& c = s->llama.cfg;
    DecRun& run = s->run;
    run.active = false;
    if (R == 0) { cz_set_error("reprime_interval must be > 0"); return CZ_ERR_ARG; }
    if (bos >= (uint32_t)c.vocab) { cz_set_error("bos id %u >= vocab", bos); return CZ_ERR_ARG; }
    size_t max_tc_in = 0;
    for (int b = 0; b < B; ++b) max_tc_in = std::max(max_tc_in, tcs[b]);
    size_t ec = context_window > 0 ? context_window - 1 : 0; if (ec > 511) ec = 511;         // :2506
    // longest context this run reaches (the schedule is known up front): it sizes the KV cache and the attention kernel's LDS inside the
    // captured graph, and with it how many blocks of other lanes' kernels fit on a CU beside the attention blocks
    size_t mx = 1;
    {
        size_t ip = 1;
        for (size_t i = 0; i < max_tc_in; ++i) {
            if (ip >= ec && (i % R) == 0 && i > 0) ip = i + 1 > ec ? ec : i + 1;
            if (ip > mx) mx = ip;
            ip += 1;
        }
    }
    if (mx >= cz_ctx_limit(s)) {                       // (mx = the last position a step runs at)
        cz_set_error("the re-prime schedule (context %u, interval %u, %zu tokens) reaches position %zu: beyond max_position_embeddings = %d (the reference fails there too) / the kernels' %d",
                     context_window, R, max_tc_in, mx, c.max_pos, CZ_KERNEL_MAX_CTX);
        return CZ_ERR_STATE;
    }
    CZCHK(ensure_streams(s, B, std::min(mx + 2, cz_ctx_limit(s))));
    CZCHK(cz_dec_run_setup(s, B, payloads, lens, caps, tcs));
    const size_t max_tc = run.max_tc;
    cz_dec_state* dst = (cz_dec_state*)s->decst.p;
    HIPCHK(cz_launch_dec_init(s->stream, dst, B));                                           // ArithmeticDecoder::new (:2485)
    hipLaunchKernelGGL(init_out_ids_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, dst, B, bos);
    CZCHK(s->tok.ensure((size_t)B * std::max<size_t>(ec, 64) * 4));
    // the captured step graph holds raw pointers: size the buffers of every later re-prime before it is built
    if (max_tc > R && ec > 0) CZCHK(reserve_rows(s, (size_t)B * ec, (size_t)B));
    hipLaunchKernelGGL(fill_u32_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, (uint32_t*)s->tok.p, bos, B);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, s->step_sym, bos, B);   // finished streams keep a valid id
    s->index_pos = 0;                                                                        // the single-stream step state is reset
    CZCHK(llama_step(s, B, 0));                                                              // step(bos) (:2485)
    // graph replay unless per-kernel profiling of the decode families is requested or the graph path is disabled
    const unsigned dec_kinds = (1u << PK_DECODE) | (1u << PK_DEC_GEMM) | (1u << PK_DEC_GEMM_QKV) | (1u << PK_DEC_GEMM_GU) | (1u << PK_DEC_GEMM_HEAD) | (1u << PK_DEC_ATTN) | (1u << PK_DEC_NORM) | (1u << PK_EMBED);
    const bool prof_decode = s->prof_on && (s->prof_mask & dec_kinds);
    const bool use_graph = !s->no_graph && !s->decode_v1 && !s->serial_cdf && !prof_decode && c.n_heads / c.n_kv_heads <= 4;
    const int max_nk = (int)std::min<size_t>(mx + 2, s->ctx_cap);
    if (use_graph && (s->graph_B != B || s->graph_gen != s->devbuf_gen || s->graph_max_nk != max_nk)) {
        CZCHK(build_step_graph(s, B, max_nk));
        if (s->graph_gen != s->devbuf_gen) CZCHK(build_step_graph(s, B, max_nk));            // the dry run itself grew a buffer: capture again
        CZCHK(llama_step(s, B, 0));                                                          // the dry run clobbered the BOS logits
    }
    hipLaunchKernelGGL(set_counters_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev, 1, 0ull);
    run.i = 0; run.index_pos = 1; run.ec = ec; run.R = R; run.bos = bos; run.use_graph = use_graph; run.max_nk = max_nk; run.active = true;
    return CZ_OK;
}
// up to n more tokens of every stream of the group (streams shorter than the longest simply finish earlier)
static cz_status decode_llama_advance(cz_session* s, size_t n) {
    const cz_llama_config& c = s->llama.cfg;
    DecRun& run = s->run;
    if (!run.active) { cz_set_error("no decode in progress on this session (cz_decode_begin_batch first)"); return CZ_ERR_STATE; }
    const int B = run.B; const size_t R = run.R, ec = run.ec;
    cz_dec_state* dst = (cz_dec_state*)s->decst.p;
    const size_t stop = run.max_tc - run.i < n ? run.max_tc : run.i + n;
    if (run.use_graph && run.i < stop && run.index_pos < s->ctx_cap && (s->graph_gen != s->devbuf_gen || s->graph_B != B || s->graph_max_nk != run.max_nk)) {
        // (only when a step will run, and never with the position at the end of the cache: the dry run inside build_step_graph writes K/V at the live position)
        // another call on this session (an encode between two slices) grew a work buffer: the captured graph holds stale pointers.
        // The step's own buffers are re-sized by the dry run inside build_step_graph; the logits of the pending step live in step_logits
        // (never moved), so re-capturing here loses nothing.
        std::vector<float> keep((size_t)B * s->llama.cfg.vocab);
        HIPCHK(hipMemcpyAsync(keep.data(), s->step_logits, keep.size() * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        CZCHK(build_step_graph(s, B, run.max_nk));
        if (s->graph_gen != s->devbuf_gen) CZCHK(build_step_graph(s, B, run.max_nk));
        HIPCHK(hipMemcpyAsync(s->step_logits, keep.data(), keep.size() * 4, hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(set_counters_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev, (int)run.index_pos, (unsigned long long)run.i);
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    for (size_t i = run.i; i < stop; ++i) {
        size_t& index_pos = run.index_pos;
        if (index_pos >= ec && (i % R) == 0 && i > 0) {                                      // :2532
            const size_t end = i + 1, hl = end > ec ? ec : end;                              // out_tokens.len() == i + 1
            if (hl == 0) { cz_set_error("reprime called with empty history"); run.active = false; return CZ_ERR_STATE; }
            hipLaunchKernelGGL(gather_hist_kernel, dim3((unsigned)((B * hl + 255) / 256)), dim3(256), 0, s->stream, dst, (uint32_t*)s->tok.p, B, (int)hl,
                               (unsigned long long)end);
            CZCHK(llama_reprime(s, B, (int)hl));
            if (run.use_graph && s->graph_gen != s->devbuf_gen) { cz_set_error("internal: a work buffer moved under the captured decode graph"); run.active = false; return CZ_ERR_STATE; }
            index_pos = hl;
            hipLaunchKernelGGL(set_counters_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev, (int)index_pos, (unsigned long long)i);
        }
        if (index_pos >= s->ctx_cap) { cz_set_error("position %zu exceeds the KV cache of this run (%zu)", index_pos, s->ctx_cap); run.active = false; return CZ_ERR_STATE; }
        if (run.use_graph) {
            if (s->probe_launch && i < 64) HIPCHK(hipStreamSynchronize(s->stream));          // probe: cost of a launch into an empty queue
            timespec c0, c1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
            const auto t0 = std::chrono::steady_clock::now();
            HIPCHK(hipGraphLaunch(s->graph_exec, s->stream));                                // decode_symbol(i); step(sym) at index_pos; ++i; ++pos
            const unsigned long long dt = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
            s->graph_launch_ns += dt;
            s->graph_launch_cpu_ns += (unsigned long long)((c1.tv_sec - c0.tv_sec) * 1000000000ll + (c1.tv_nsec - c0.tv_nsec));
            if (s->probe_launch && i < 64) s->graph_launch_idle_ns += dt;
        } else {
            s->prof_begin(PK_DECODE);
            if (s->serial_cdf) HIPCHK(cz_launch_decode_symbol_serial(s->stream, s->step_logits, c.vocab, dst, s->step_sym, B, c.vocab, 0, (unsigned long long)i, nullptr));
            else HIPCHK(cz_launch_decode_symbol(s->stream, s->step_logits, c.vocab, dst, s->step_sym, B, c.vocab, 0, (unsigned long long)i, s->fallback_dev, nullptr, s->cdf_scratch));
            s->prof_end(0, (double)B * c.vocab * 4);
            if (i + 1 < run.max_tc) {
                // the reference also steps after the last symbol (:2627); its logits are never used, so it is skipped here
                HIPCHK(hipMemcpyAsync(s->tok.p, s->step_sym, (size_t)B * 4, hipMemcpyDeviceToDevice, s->stream));
                CZCHK(llama_step(s, B, (int)index_pos));
            }
            if (s->step_sync) HIPCHK(hipStreamSynchronize(s->stream));    // never more than one step of launches queued (profiler runs)
        }
        // Bound what is queued behind the host (a token is 110-220 kernel packets): a faulting kernel is reported within CZ_SYNC_EVERY tokens and
        // the host never runs thousands of packets ahead of the device.  One wait every CZ_SYNC_EVERY tokens costs one launch latency per 8
        // steps (< 0.3 % even for a single stream).  This is NOT a cure for the rocprofv3 SIGSEGV: that one is ROCr's intercept
        // queue handing the profiler a packet batch that wraps the AQL ring's end — any multi-packet doorbell can do it (every hipGraphLaunch
        // is one), a shallow queue only makes the batches of EAGER launches small.  Profile with tools/profile_pass.py (eager, step_sync).
        if (((i - run.i) % CZ_SYNC_EVERY) == CZ_SYNC_EVERY - 1) HIPCHK(hipStreamSynchronize(s->stream));
        index_pos += 1;
    }
    run.i = stop;
    HIPCHK(hipStreamSynchronize(s->stream));       // the slice is done when the call returns (and a faulting kernel is reported by this call)
    return CZ_OK;
}
static cz_status decode_llama_group(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* tcs,
                                    uint32_t context_window, uint32_t R, uint32_t bos, uint32_t* const* out_ids) {
    CZCHK(decode_llama_begin(s, B, payloads, lens, nullptr, tcs, context_window, R, bos));
    cz_status rc = decode_llama_advance(s, s->run.max_tc);
    if (rc == CZ_OK) rc = cz_dec_run_fetch(s, out_ids);
    s->run.active = false;
    return rc;
}

// ------------------------------------------------------------------ C ABI
static inline bool cz_dead(const cz_session* s) {
    if (s && s->orphan) { cz_set_error("this session is a fork whose parent (the owner of the weights) was closed"); return true; }
    return false;
}
extern "C" cz_status cz_session_open(const cz_model_desc* desc, int device_ordinal, cz_session** out) {
    if (!desc || !out) { cz_set_error("cz_session_open: null argument"); return CZ_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { cz_set_error("no HIP device available (this library has no CPU fallback)"); return CZ_ERR_HIP; }
    if (device_ordinal < 0 || device_ordinal >= ndev) { cz_set_error("device ordinal %d out of range (%d devices)", device_ordinal, ndev); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { cz_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device_ordinal, prop.gcnArchName); return CZ_ERR_HIP; }
    cz_session* s = new cz_session();
    s->backend = desc->backend; s->device = device_ordinal; s->seed = desc->synthetic_seed; s->round_f32 = desc->round_f32_weights_to_bf16 != 0;
    s->max_streams = desc->max_streams ? (int)desc->max_streams : 64;
    if (s->max_streams > CZ_CDF_PAR_MAX_GRID) { delete s; cz_set_error("max_streams > %d is not supported", CZ_CDF_PAR_MAX_GRID); return CZ_ERR_ARG; }
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);   // several sessions may capture/replay graphs concurrently
    if (e != hipSuccess) { delete s; cz_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return CZ_ERR_HIP; }
    if (hipMalloc((void**)&s->fallback_dev, 64) != hipSuccess || hipMemset(s->fallback_dev, 0, 64) != hipSuccess) { delete s; cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    cz_status rc = CZ_OK;
    if (desc->backend == CZ_BACKEND_SMOLLM) rc = build_llama(s, desc);
    else if (desc->backend == CZ_BACKEND_RWKV7) rc = cz_rwkv7_build(s, desc);
    else { cz_set_error("unknown backend %d", (int)desc->backend); rc = CZ_ERR_ARG; }      // src/main.rs:1829 "unknown backend"
    if (rc == CZ_OK && desc->backend == CZ_BACKEND_SMOLLM && cz_cdf_par_supported(s->llama.cfg.vocab, 0) &&
        hipMalloc((void**)&s->cdf_scratch, cz_cdf_par_scratch_bytes(s->llama.cfg.vocab, 0)) != hipSuccess) { cz_set_error("cannot allocate CDF scratch"); rc = CZ_ERR_HIP; }
    if (rc == CZ_OK && desc->backend == CZ_BACKEND_RWKV7 && cz_cdf_par_supported((int)cz_rwkv7_vocab(s), 1) &&
        hipMalloc((void**)&s->cdf_scratch, cz_cdf_par_scratch_bytes((int)cz_rwkv7_vocab(s), 1)) != hipSuccess) { cz_set_error("cannot allocate CDF scratch"); rc = CZ_ERR_HIP; }
    if (rc == CZ_OK && cz_dec_attn_prepare(CZ_DEF_CTX) != hipSuccess) { cz_set_error("cannot reserve LDS for decode attention"); rc = CZ_ERR_HIP; }
    if (rc == CZ_OK && cz_attention_prepare(CZ_DEF_CTX) != hipSuccess) { cz_set_error("cannot reserve LDS for attention"); rc = CZ_ERR_HIP; }
    if (rc == CZ_OK && (cz_gemm_prepare() != hipSuccess || cz_gemm_mid_prepare() != hipSuccess || cz_dec_gemm_prepare() != hipSuccess)) { cz_set_error("cannot reserve LDS for the GEMM kernels"); rc = CZ_ERR_HIP; }
    if (rc != CZ_OK) { cz_session_close(s); return rc; }
    *out = s;
    return CZ_OK;
}
extern "C" void cz_session_close(cz_session* s) {
    if (!s) return;
    if (s->fork_of) { auto& v = s->fork_of->forks; v.erase(std::remove(v.begin(), v.end(), s), v.end()); }
    for (cz_session* f : s->forks) {                                // forks outlive their parent only as husks: no weights, every call fails
        f->fork_of = nullptr; f->orphan = true;
        f->llama.blob = f->llama.blob_dec = f->llama.blob_gu = f->llama.blob_k2 = nullptr; f->llama.rope_cos = f->llama.rope_sin = nullptr;
    }
    s->forks.clear();
    hipSetDevice(s->device);
    for (cz_session* l : s->lanes) cz_session_close(l);
    s->lanes.clear();
    if (s->is_lane) { s->llama.blob = s->llama.blob_dec = s->llama.blob_gu = s->llama.blob_k2 = nullptr; s->llama.rope_cos = s->llama.rope_sin = nullptr; }   // the weights belong to the parent
    if (s->stream) hipStreamSynchronize(s->stream);
    destroy_step_graph(s);
    if (s->dbg_ge) hipGraphExecDestroy(s->dbg_ge);
    if (s->dbg_g) hipGraphDestroy(s->dbg_g);
    s->prof_drain();
    for (auto e : s->evt_free) hipEventDestroy(e);
    for (DevBuf* b : s->all_bufs()) b->release();
    void* ptrs[] = {s->cdf_scratch, s->fallback_dev, s->llama.blob_gu, s->llama.blob_k2, s->llama.blob_dec, s->llama.blob, s->llama.rope_cos, s->llama.rope_sin, s->kvcache, s->pos_dev, s->step_seqd, s->step_roped, s->step_rows,
                    s->step_logits, s->step_sym, s->attn_split};
    for (auto p : ptrs) if (p) { hipError_t e = hipFree(p); (void)e; }
    cz_rwkv7_destroy(s);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}
extern "C" size_t cz_vocab_size(const cz_session* s) { return s ? (s->backend == CZ_BACKEND_SMOLLM ? (size_t)s->llama.cfg.vocab : cz_rwkv7_vocab(s)) : 0; }
extern "C" size_t cz_max_context_length(const cz_session* s) { return s && s->backend == CZ_BACKEND_SMOLLM ? 512 : (size_t)-1; }   // src/models.rs:91, :149
extern "C" size_t cz_index_pos(const cz_session* s) { return s ? s->index_pos : 0; }

extern "C" cz_status cz_step_logits(cz_session* s, uint32_t token, float* out_logits) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !out_logits) { cz_set_error("cz_step_logits: null argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (s->backend == CZ_BACKEND_RWKV7) return cz_rwkv7_step_host(s, token, out_logits);
    if (token >= (uint32_t)s->llama.cfg.vocab) { cz_set_error("token id %u >= vocab %d", token, s->llama.cfg.vocab); return CZ_ERR_ARG; }
    CZCHK(ensure_streams(s, 1, s->index_pos + 1, true));        // (a stream that steps past its cache keeps its keys: the cache is re-laid out longer)
    CZCHK(s->tok.ensure(64));
    HIPCHK(hipMemcpyAsync(s->tok.p, &token, 4, hipMemcpyHostToDevice, s->stream));
    CZCHK(llama_step(s, 1, (int)s->index_pos));
    HIPCHK(hipMemcpyAsync(out_logits, s->step_logits, (size_t)s->llama.cfg.vocab * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->index_pos += 1;                                                                      // src/models.rs:95
    return CZ_OK;
}
extern "C" cz_status cz_reprime_logits(cz_session* s, const uint32_t* history, size_t n, float* out_logits) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !out_logits) { cz_set_error("cz_reprime_logits: null argument"); return CZ_ERR_ARG; }
    if (!history || n == 0) { cz_set_error("reprime called with empty history"); return CZ_ERR_ARG; }   // src/models.rs:108 / :164
    HIPCHK(hipSetDevice(s->device));
    if (s->backend == CZ_BACKEND_RWKV7) return cz_rwkv7_reprime_host(s, history, n, out_logits);
    for (size_t i = 0; i < n; ++i) if (history[i] >= (uint32_t)s->llama.cfg.vocab) { cz_set_error("token id %u >= vocab", history[i]); return CZ_ERR_ARG; }
    if (n > cz_ctx_limit(s)) { cz_set_error("history of %zu tokens exceeds max_position_embeddings = %d / the kernels' %d", n, s->llama.cfg.max_pos, CZ_KERNEL_MAX_CTX); return CZ_ERR_STATE; }
    CZCHK(ensure_streams(s, 1, n));
    CZCHK(s->tok.ensure(n * 4));
    HIPCHK(hipMemcpyAsync(s->tok.p, history, n * 4, hipMemcpyHostToDevice, s->stream));
    s->index_pos = 0;                                                                       // fresh cache (src/models.rs:105-107)
    CZCHK(llama_reprime(s, 1, (int)n));
    HIPCHK(hipMemcpyAsync(out_logits, s->step_logits, (size_t)s->llama.cfg.vocab * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->index_pos = n;                                                                       // src/models.rs:111
    return CZ_OK;
}

// ---- decode lanes.  One lock-step group is latency-bound (about sixty small dependent kernels per token), so groups beyond
// the first run concurrently on sessions of their own (stream, KV cache, work buffers, step graph) that share the weights.
static cz_status make_lane(cz_session* p, cz_session** out) {
    cz_session* c = new cz_session();
    c->is_lane = true; c->backend = p->backend; c->device = p->device; c->seed = p->seed; c->max_streams = p->max_streams;
    c->llama = p->llama;                                   // shallow: same device tensors
    c->max_lanes = 1;
    const bool rw = p->backend == CZ_BACKEND_RWKV7;
    if (rw && cz_rwkv7_make_lane(p, c) != CZ_OK) { cz_session_close(c); return CZ_ERR_HIP; }
    const size_t scratch = rw ? cz_cdf_par_scratch_bytes((int)cz_rwkv7_vocab(p), 1) : cz_cdf_par_scratch_bytes(c->llama.cfg.vocab, 0);
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, p->stream_prio) != hipSuccess || hipMalloc((void**)&c->fallback_dev, 64) != hipSuccess ||
        hipMemset(c->fallback_dev, 0, 64) != hipSuccess || (p->cdf_scratch && hipMalloc((void**)&c->cdf_scratch, scratch) != hipSuccess)) {
        cz_session_close(c); cz_set_error("cannot allocate a lane"); return CZ_ERR_HIP;
    }
    *out = c;
    return CZ_OK;
}
static void lane_inherit(const cz_session* s, cz_session* ls) {   // a lane runs with its parent's knobs
    ls->kv_ctx_cap = s->kv_ctx_cap; ls->serial_cdf = s->serial_cdf; ls->decode_v1 = s->decode_v1; ls->no_graph = s->no_graph; ls->force_big_gemm = s->force_big_gemm;
    ls->max_batch_rows = s->max_batch_rows; ls->logit_batch_rows = s->logit_batch_rows; ls->step_sync = s->step_sync; ls->fuse_norm = s->fuse_norm; ls->wide_min_streams = s->wide_min_streams; ls->wide_mode = s->wide_mode; ls->dbg_skip = s->dbg_skip; ls->no_last_prune = s->no_last_prune; ls->rwkv_seq = s->rwkv_seq; ls->rwkv_wide_min = s->rwkv_wide_min; ls->rwkv_wide_mode = s->rwkv_wide_mode;
}
// run `n_streams` streams as groups of <= max_streams on up to max_lanes lanes; group(ls, off, count) processes streams [off, off + count)
template <class GroupFn>
static cz_status run_on_lanes(cz_session* s, size_t n_streams, GroupFn group) {
    const size_t cap = (size_t)s->max_streams;
    const size_t G = (n_streams + cap - 1) / cap;
    // groups of near-equal size, group g = streams [off[g], off[g+1])
    std::vector<size_t> off(G + 1, 0);
    for (size_t g = 0; g < G; ++g) off[g + 1] = off[g] + n_streams / G + (g < n_streams % G ? 1 : 0);
    const size_t L = std::min(G, (size_t)std::max(1, s->max_lanes));
    while (s->lanes.size() + 1 < L) { cz_session* c = nullptr; CZCHK(make_lane(s, &c)); s->lanes.push_back(c); }
    std::vector<cz_status> rc(L, CZ_OK); std::vector<std::string> msg(L);
    auto run = [&](size_t l) {
        cz_session* ls = l == 0 ? s : s->lanes[l - 1];
        if (hipSetDevice(s->device) != hipSuccess) { rc[l] = CZ_ERR_HIP; msg[l] = "hipSetDevice failed"; return; }
        if (l) lane_inherit(s, ls);
        ls->lanes_concurrent = (int)L;
        for (size_t g = l; g < G && rc[l] == CZ_OK; g += L) {
            rc[l] = group(ls, off[g], off[g + 1] - off[g]);
            if (rc[l] != CZ_OK) msg[l] = cz_last_error();
        }
    };
    std::vector<std::thread> th;
    for (size_t l = 1; l < L; ++l) th.emplace_back(run, l);
    run(0);
    for (auto& t : th) t.join();
    for (size_t l = 0; l < L; ++l) if (rc[l] != CZ_OK) { cz_set_error("%s", msg[l].c_str()); return rc[l]; }
    return CZ_OK;
}

extern "C" cz_status cz_encode_tokens_batch(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n, uint32_t context,
                                            uint32_t reprime_interval, uint32_t* const* out_c_lo, uint32_t* const* out_c_hi) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !ids || !n || !out_c_lo || !out_c_hi) { cz_set_error("cz_encode_tokens_batch: null argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (s->backend == CZ_BACKEND_RWKV7) {                  // RWKV has no re-prime: encode walks the recurrence in lock-step too, one group per lane
        if (n_streams == 0) return CZ_OK;
        if (s->rwkv_seq) return cz_rwkv7_encode_seq(s, n_streams, ids, n, out_c_lo, out_c_hi);   // teacher-forced: all tokens of a block through the GEMMs at once
        return run_on_lanes(s, n_streams, [&](cz_session* ls, size_t o, size_t cnt) {
            return cz_rwkv7_encode_group(ls, (int)cnt, ids + o, n + o, out_c_lo + o, out_c_hi + o);
        });
    }
    return encode_llama(s, n_streams, ids, n, context, reprime_interval, nullptr, nullptr, out_c_lo, out_c_hi);
}
extern "C" cz_status cz_encode_tokens(cz_session* s, const uint32_t* ids, size_t n, uint32_t context, uint32_t reprime_interval,
                                      uint32_t* out_c_lo, uint32_t* out_c_hi) {
    return cz_encode_tokens_batch(s, 1, &ids, &n, context, reprime_interval, &out_c_lo, &out_c_hi);
}
extern "C" cz_status cz_decode_tokens_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens,
                                            const size_t* token_counts, uint32_t context_window, uint32_t reprime_interval, uint32_t bos,
                                            uint32_t* const* out_ids) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !payloads || !lens || !token_counts || !out_ids) { cz_set_error("cz_decode_tokens_batch: null argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (n_streams == 0) return CZ_OK;
    const bool rw = s->backend == CZ_BACKEND_RWKV7;
    return run_on_lanes(s, n_streams, [&](cz_session* ls, size_t o, size_t cnt) {
        return rw ? cz_rwkv7_decode_group(ls, (int)cnt, payloads + o, lens + o, token_counts + o, bos, out_ids + o)
                  : decode_llama_group(ls, (int)cnt, payloads + o, lens + o, token_counts + o, context_window, reprime_interval, bos, out_ids + o);
    });
}
extern "C" cz_status cz_decode_tokens(cz_session* s, const uint8_t* payload, size_t len, size_t token_count, uint32_t context_window,
                                      uint32_t reprime_interval, uint32_t bos, uint32_t* out_ids) {
    return cz_decode_tokens_batch(s, 1, &payload, &len, &token_count, context_window, reprime_interval, bos, &out_ids);
}

// ---- the decode loop in slices.  Every lock-step group stays resident on a lane of its own between the calls, so the streams must fit
// max_streams x max_lanes.  group(ls, g) runs on lane g's host thread.
template <class GroupFn>
static cz_status on_run_groups(cz_session* s, GroupFn group) {
    const size_t G = s->run_off.size() > 1 ? s->run_off.size() - 1 : 0;
    if (G == 0) { cz_set_error("no decode in progress on this session (cz_decode_begin_batch first)"); return CZ_ERR_STATE; }
    std::vector<cz_status> rc(G, CZ_OK); std::vector<std::string> msg(G);
    auto run = [&](size_t g) {
        cz_session* ls = g == 0 ? s : s->lanes[g - 1];
        if (hipSetDevice(s->device) != hipSuccess) { rc[g] = CZ_ERR_HIP; msg[g] = "hipSetDevice failed"; return; }
        if (g) lane_inherit(s, ls);
        ls->lanes_concurrent = (int)G;
        rc[g] = group(ls, g);
        if (rc[g] != CZ_OK) msg[g] = cz_last_error();
    };
    std::vector<std::thread> th;
    for (size_t g = 1; g < G; ++g) th.emplace_back(run, g);
    run(0);
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; ++g) if (rc[g] != CZ_OK) { cz_set_error("%s", msg[g].c_str()); return rc[g]; }
    return CZ_OK;
}
static cz_status decode_begin_any(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* token_counts,
                                  uint32_t context_window, uint32_t reprime_interval, uint32_t bos) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !payloads || !lens || !token_counts || n_streams == 0) { cz_set_error("cz_decode_begin_batch: null argument / no streams"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    s->run_off.clear();
    const size_t cap = (size_t)s->max_streams, G = (n_streams + cap - 1) / cap;
    if (G > (size_t)std::max(1, s->max_lanes)) {
        cz_set_error("a sliced decode keeps every lock-step group resident: %zu streams need %zu lanes of %d streams, max_lanes is %d", n_streams, G, s->max_streams, s->max_lanes);
        return CZ_ERR_UNSUPPORTED;
    }
    while (s->lanes.size() + 1 < G) { cz_session* c = nullptr; CZCHK(make_lane(s, &c)); s->lanes.push_back(c); }
    s->run_off.assign(G + 1, 0);
    for (size_t g = 0; g < G; ++g) s->run_off[g + 1] = s->run_off[g] + n_streams / G + (g < n_streams % G ? 1 : 0);
    const bool rw = s->backend == CZ_BACKEND_RWKV7;
    const cz_status rc = on_run_groups(s, [&](cz_session* ls, size_t g) {
        const size_t o = s->run_off[g]; const int cnt = (int)(s->run_off[g + 1] - o);
        return rw ? cz_rwkv7_decode_begin(ls, cnt, payloads + o, lens + o, caps ? caps + o : nullptr, token_counts + o, bos)
                  : decode_llama_begin(ls, cnt, payloads + o, lens + o, caps ? caps + o : nullptr, token_counts + o, context_window, reprime_interval, bos);
    });
    if (rc != CZ_OK) s->run_off.clear();
    return rc;
}
extern "C" cz_status cz_decode_begin_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens, const size_t* token_counts,
                                           uint32_t context_window, uint32_t reprime_interval, uint32_t bos) {
    return decode_begin_any(s, n_streams, payloads, lens, nullptr, token_counts, context_window, reprime_interval, bos);
}
extern "C" cz_status cz_decode_begin_partial_batch(cz_session* s, size_t n_streams, const uint8_t* const* payloads, const size_t* lens, const size_t* payload_caps,
                                                   const size_t* token_counts, uint32_t context_window, uint32_t reprime_interval, uint32_t bos) {
    if (!payload_caps) { cz_set_error("cz_decode_begin_partial_batch: null capacities"); return CZ_ERR_ARG; }
    return decode_begin_any(s, n_streams, payloads, lens, payload_caps, token_counts, context_window, reprime_interval, bos);
}
extern "C" cz_status cz_decode_feed_batch(cz_session* s, size_t n_streams, const uint8_t* const* chunks, const size_t* lens, int final_input) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !chunks || !lens) { cz_set_error("cz_decode_feed_batch: null argument"); return CZ_ERR_ARG; }
    if (s->run_off.empty() || n_streams != s->run_off.back()) { cz_set_error("cz_decode_feed_batch: %zu streams, the decode in progress has %zu", n_streams, s->run_off.empty() ? (size_t)0 : s->run_off.back()); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    return on_run_groups(s, [&](cz_session* ls, size_t g) { return cz_dec_run_feed(ls, chunks + s->run_off[g], lens + s->run_off[g], final_input); });
}
extern "C" cz_status cz_decode_advance(cz_session* s, size_t n_tokens, size_t* position, size_t* remaining) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s) { cz_set_error("cz_decode_advance: null argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    const bool rw = s->backend == CZ_BACKEND_RWKV7;
    CZCHK(on_run_groups(s, [&](cz_session* ls, size_t) { return rw ? cz_rwkv7_decode_advance(ls, n_tokens) : decode_llama_advance(ls, n_tokens); }));
    size_t pos = 0, rem = 0;
    for (size_t g = 0; g + 1 < s->run_off.size(); ++g) {
        const DecRun& r = (g == 0 ? s : s->lanes[g - 1])->run;
        pos = std::max(pos, r.i); rem = std::max(rem, r.max_tc - r.i);
    }
    if (position) *position = pos;
    if (remaining) *remaining = rem;
    return CZ_OK;
}
extern "C" cz_status cz_decode_fetch_batch(cz_session* s, size_t n_streams, uint32_t* const* out_ids) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !out_ids) { cz_set_error("cz_decode_fetch_batch: null argument"); return CZ_ERR_ARG; }
    if (s->run_off.empty() || n_streams != s->run_off.back()) { cz_set_error("cz_decode_fetch_batch: %zu streams, the decode in progress has %zu", n_streams, s->run_off.empty() ? (size_t)0 : s->run_off.back()); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    return on_run_groups(s, [&](cz_session* ls, size_t g) { return cz_dec_run_fetch(ls, out_ids + s->run_off[g]); });
}
extern "C" cz_status cz_decode_end(cz_session* s) {
    if (!s) return CZ_ERR_ARG;
    for (size_t g = 0; g + 1 < s->run_off.size(); ++g) (g == 0 ? s : s->lanes[g - 1])->run.active = false;
    s->run_off.clear();
    return CZ_OK;
}
extern "C" cz_status cz_encode_tokens_range_batch(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n, uint32_t context,
                                                  uint32_t reprime_interval, const size_t* i_begin, const size_t* i_end,
                                                  uint32_t* const* out_c_lo, uint32_t* const* out_c_hi) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !ids || !n || !i_begin || !i_end || !out_c_lo || !out_c_hi) { cz_set_error("cz_encode_tokens_range_batch: null argument"); return CZ_ERR_ARG; }
    if (s->backend != CZ_BACKEND_SMOLLM) {
        cz_set_error("cz_encode_tokens_range_batch: the rwkv7 backend has no re-prime (src/main.rs:2275), so a stream's tokens cannot be evaluated independently");
        return CZ_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(s->device));
    return encode_llama(s, n_streams, ids, n, context, reprime_interval, i_begin, i_end, out_c_lo, out_c_hi);
}

// a second, independent session on the SAME device tensors (the reference loads the model a second time for that, src/main.rs:1943-1956)
extern "C" cz_status cz_session_fork(cz_session* parent, cz_session** out) {
    if (cz_dead(parent)) return CZ_ERR_STATE;
    if (!parent || !out) { cz_set_error("cz_session_fork: null argument"); return CZ_ERR_ARG; }
    *out = nullptr;
    if (parent->is_lane) { cz_set_error("cz_session_fork: fork the original session, not a fork"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(parent->device));
    cz_session* c = nullptr;
    CZCHK(make_lane(parent, &c));
    c->max_lanes = parent->max_lanes;                       // a fork may run its own lanes
    c->fork_of = parent; parent->forks.push_back(c);
    *out = c;
    return CZ_OK;
}
extern "C" cz_status cz_xe_bits_batch(cz_session* s, size_t n_spans, const uint32_t* const* history, const size_t* n_history, const uint32_t* const* hint,
                                      const size_t* n_hint, const uint32_t* const* targets, const size_t* n_targets, double* bits_out) {
    if (cz_dead(s)) return CZ_ERR_STATE;
    if (!s || !targets || !n_targets || !bits_out || (n_history && !history) || (n_hint && !hint)) { cz_set_error("cz_xe_bits_batch: null argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (s->backend == CZ_BACKEND_RWKV7) return cz_rwkv7_xe_bits(s, n_spans, history, n_history, hint, n_hint, targets, n_targets, bits_out);   // cross_entropy_bits_over_span_rwkv
    return xe_llama(s, n_spans, history, n_history, hint, n_hint, targets, n_targets, bits_out);
}
extern "C" cz_status cz_xe_bits_over_span(cz_session* s, const uint32_t* history, size_t n_history, const uint32_t* hint, size_t n_hint,
                                          const uint32_t* targets, size_t n_targets, double* bits_out) {
    return cz_xe_bits_batch(s, 1, &history, &n_history, &hint, &n_hint, &targets, &n_targets, bits_out);
}

extern "C" cz_status cz_profile_enable(cz_session* s, int on) {
    if (!s) return CZ_ERR_ARG;
    s->prof_drain();
    for (int k = 0; k < PK_COUNT; ++k) s->acc[k] = ProfAcc();
    s->prof_on = on != 0;
    s->prof_mask = on > 1 ? (unsigned)on >> 1 : 0xFFFFFFFFu;      // on = 1: every kind; on = 1 | (mask << 1): selected kinds
    return CZ_OK;
}
extern "C" cz_status cz_profile_read(cz_session* s, cz_kernel_stat* out, size_t cap, size_t* n) {
    if (!s || !n) return CZ_ERR_ARG;
    s->prof_drain();
    size_t k = 0;
    for (int i = 0; i < PK_COUNT; ++i) {
        if (!s->acc[i].launches) continue;
        if (k < cap && out) {
            memset(&out[k], 0, sizeof out[k]);
            snprintf(out[k].name, sizeof out[k].name, "%s", kProfNames[i]);
            out[k].launches = s->acc[i].launches; out[k].total_ms = s->acc[i].ms; out[k].flops = s->acc[i].flops; out[k].bytes = s->acc[i].bytes;
        }
        ++k;
    }
    *n = k;
    return CZ_OK;
}
// the streams THIS session's work runs on — its own, its decode lanes', its forks' (and their lanes') — and no other stream of the device:
// another session, torch or RCCL work of the same process is not waited for
static cz_status sync_session_tree(cz_session* s) {
    if (s->stream) HIPCHK(hipStreamSynchronize(s->stream));
    for (cz_session* l : s->lanes) if (l) CZCHK(sync_session_tree(l));
    for (cz_session* f : s->forks) if (f) CZCHK(sync_session_tree(f));
    return CZ_OK;
}
extern "C" cz_status cz_synchronize(cz_session* s) {
    if (!s) return CZ_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    return sync_session_tree(s);
}
// tuning aid (tools/bench_decode_step.py): lock-step decode steps of B streams at a FIXED position `pos` (KV cache zeroed), as a replayed
// hipGraph (use_graph) or eagerly (per-family HIP events when profiling is on).  iters <= 0 prepares (buffers, graph capture, one warm
// step) so that several lanes can start their timed loops together; iters > 0 runs the loop: *ms_total = device time of the iters steps.
extern "C" cz_status cz_debug_decode_steps(cz_session* s, int B, int pos, int iters, int use_graph, int with_symbol, double* ms_total) {
    if (!s || !ms_total || s->backend != CZ_BACKEND_SMOLLM || B <= 0) { cz_set_error("cz_debug_decode_steps: bad argument"); return CZ_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    const cz_llama_config& c = s->llama.cfg;
    auto one = [&]() -> cz_status {
        if (with_symbol) HIPCHK(cz_launch_decode_symbol(s->stream, s->step_logits, c.vocab, (cz_dec_state*)s->decst.p, s->step_sym, B, c.vocab, 0, 0ull, s->fallback_dev, nullptr, s->cdf_scratch));
        return llama_step_any(s, B, pos, use_graph ? (pos + 1 <= 1024 ? 1024 : (pos + 1 <= CZ_DEF_CTX ? CZ_DEF_CTX : (int)s->ctx_cap)) : pos + 1);
    };
    if (iters <= 0) {
        CZCHK(ensure_streams(s, B, pos >= 0 ? (size_t)pos + 1 : 0));
        if (pos < 0 || pos >= (int)s->ctx_cap) { cz_set_error("pos out of range"); return CZ_ERR_ARG; }
        CZCHK(s->tok.ensure((size_t)B * 4)); CZCHK(s->decst.ensure(B * sizeof(cz_dec_state))); CZCHK(s->payload.ensure((size_t)B * 64)); CZCHK(s->outids.ensure((size_t)B * 16));
        HIPCHK(hipMemsetAsync(s->kvcache, 0, (size_t)s->max_streams * s->kv_stream_stride * 4, s->stream));
        HIPCHK(hipMemsetAsync(s->payload.p, 0x5a, (size_t)B * 64, s->stream));
        std::vector<cz_dec_state> hs(B);
        for (int b = 0; b < B; ++b) {
            cz_dec_state d; memset(&d, 0, sizeof d);
            d.in = (const uint8_t*)s->payload.p + (size_t)b * 64; d.in_len = 64; d.out_ids = (uint32_t*)s->outids.p + (size_t)b * 4; d.token_count = 1u << 30; d.active = 1; d.final_input = 1;
            hs[b] = d;
        }
        HIPCHK(hipMemcpyAsync(s->decst.p, hs.data(), B * sizeof(cz_dec_state), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(cz_launch_dec_init(s->stream, (cz_dec_state*)s->decst.p, B));
        hipLaunchKernelGGL(fill_u32_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, (uint32_t*)s->tok.p, 1u, B);
        hipLaunchKernelGGL(set_counters_kernel, dim3(1), dim3(1), 0, s->stream, s->pos_dev, pos, 0ull);    // the symbol kernel writes out_ids[1] every step (step index 0)
        CZCHK(one());                                                   // sizes every buffer, pages the code in
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->dbg_ge) { hipGraphExecDestroy(s->dbg_ge); s->dbg_ge = nullptr; }
        if (s->dbg_g) { hipGraphDestroy(s->dbg_g); s->dbg_g = nullptr; }
        if (use_graph) {
            const bool prof = s->prof_on; s->prof_on = false;
            HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
            const cz_status rc = one();
            const hipError_t e2 = hipStreamEndCapture(s->stream, &s->dbg_g);
            s->prof_on = prof;
            if (rc != CZ_OK || e2 != hipSuccess || !s->dbg_g) { cz_set_error("capture failed"); return CZ_ERR_HIP; }
            HIPCHK(hipGraphInstantiate(&s->dbg_ge, s->dbg_g, nullptr, nullptr, 0));
            HIPCHK(hipGraphLaunch(s->dbg_ge, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        *ms_total = 0.0;
        return CZ_OK;
    }
    if (use_graph && !s->dbg_ge) { cz_set_error("cz_debug_decode_steps: prepare first (iters <= 0)"); return CZ_ERR_STATE; }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, s->stream));
    for (int i = 0; i < iters; ++i) {
        if (use_graph) HIPCHK(hipGraphLaunch(s->dbg_ge, s->stream)); else CZCHK(one());
        if (s->step_sync) HIPCHK(hipStreamSynchronize(s->stream));      // (profiler runs: one doorbell per packet, never a backlog — DESIGN.md §9)
    }
    HIPCHK(hipEventRecord(e1, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_total = ms;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return CZ_OK;
}
// restrict this session's stream to the CUs of one XCD (experiment: lanes that do not share an L2).  CU-mask bit i belongs to XCD i % 8.
static cz_status set_stream_xcd(cz_session* s, int xcd, int n_xcd) {
    if (s->kvcache || s->graph_exec) { cz_set_error("cu_mask_xcd must be set before the first step / decode"); return CZ_ERR_STATE; }
    hipStream_t ns = nullptr;
    if (xcd < 0) HIPCHK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    else {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i) { const int x = i % 8; if (x >= xcd && x < xcd + n_xcd) mask[i / 32] |= 1u << (i % 32); }
        HIPCHK(hipExtStreamCreateWithCUMask(&ns, 8, mask));
    }
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    s->stream = ns;
    return CZ_OK;
}

// replace this session's stream by one of another priority (HIP: lower number = higher priority; clamped to the device's range).
// For a session that shares the device with latency-bound work of other sessions, e.g. a forked encoder beside decode lanes.
static cz_status set_stream_priority(cz_session* s, int prio) {
    if (s->graph_exec || s->run.active) { cz_set_error("stream_priority must be set before the first decode"); return CZ_ERR_STATE; }
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (prio > least) prio = least;
    if (prio < greatest) prio = greatest;
    hipStream_t ns = nullptr;
    HIPCHK(hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, prio));
    if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
    s->stream = ns; s->stream_prio = prio;
    return CZ_OK;
}

// The process-wide knobs (the g_* globals of the kernel files) with the values they are shipped with, captured the first time any knob is
// touched: "reset_process_knobs" puts every one of them back (the tests' fixture finalizer: a test that turned one leaves no trace for
// the tests that run after it in the same process).
extern int g_gemv_max_rows, g_gemv_fold_attn, g_cdf_dec_grid_cap, g_gemm_lds_kb, g_gemm_pipe, g_dec_attn_waves, g_attn_v1, g_dec_grid_target, g_dec_nw_cap, g_dec_mt_max, g_dec_norm_mfma, g_gemm_mid_rows, g_mid_nw, g_mid_colmaj, g_mid_quad,
           g_dec_attn_split_heads, g_dec_attn_lds_pad, g_dec_wide_rows, g_dec_attn_small_grid, g_dec_attn_split_max, g_rw_seq_rows, g_rw_state_pad;
struct ProcKnob { const char* key; int* p; int shipped; };
static std::vector<ProcKnob>& proc_knobs() {
    static std::vector<ProcKnob> k;
    if (k.empty()) {
        const std::pair<const char*, int*> all[] = {
            {"gemv_max_rows", &g_gemv_max_rows}, {"gemv_fold_attn", &g_gemv_fold_attn}, {"cdf_dec_grid_cap", &g_cdf_dec_grid_cap}, {"gemm_lds_kb", &g_gemm_lds_kb}, {"gemm_pipe", &g_gemm_pipe}, {"dec_attn_waves", &g_dec_attn_waves}, {"attn_v1", &g_attn_v1}, {"dec_grid_target", &g_dec_grid_target}, {"dec_nw_cap", &g_dec_nw_cap},
            {"dec_mt_max", &g_dec_mt_max}, {"dec_norm_mfma", &g_dec_norm_mfma}, {"gemm_mid_rows", &g_gemm_mid_rows}, {"mid_nw", &g_mid_nw}, {"mid_colmaj", &g_mid_colmaj},
            {"mid_quad", &g_mid_quad}, {"dec_attn_split_heads", &g_dec_attn_split_heads}, {"dec_attn_lds_pad", &g_dec_attn_lds_pad},
            {"dec_wide_rows", &g_dec_wide_rows}, {"dec_attn_small_grid", &g_dec_attn_small_grid}, {"dec_attn_split_max", &g_dec_attn_split_max},
            {"rwkv_seq_rows", &g_rw_seq_rows}, {"rw_state_pad", &g_rw_state_pad}};
        for (const auto& a : all) k.push_back({a.first, a.second, *a.second});
    }
    return k;
}
// test/bench knobs (not part of the drop-in surface): 0 = default kernels
extern "C" cz_status cz_debug_set(cz_session* s, const char* key, long value) {
    if (!key) return CZ_ERR_ARG;
    std::vector<ProcKnob>& pk = proc_knobs();                // (before anything is changed: the shipped values)
    if (!strcmp(key, "reset_process_knobs")) { for (auto& k : pk) *k.p = k.shipped; return CZ_OK; }   // (no session needed)
    if (!s) return CZ_ERR_ARG;
    if (!strcmp(key, "force_big_gemm")) s->force_big_gemm = (int)value;
    else if (!strcmp(key, "max_batch_rows")) s->max_batch_rows = (size_t)value;
    else if (!strcmp(key, "logit_batch_rows")) s->logit_batch_rows = (size_t)value;
    else if (!strcmp(key, "serial_cdf")) s->serial_cdf = (int)value;
    else if (!strcmp(key, "decode_v1")) s->decode_v1 = (int)value;
    else if (!strcmp(key, "no_graph")) s->no_graph = (int)value;
    else if (!strcmp(key, "attn_v1")) g_attn_v1 = (int)value;
    else if (!strcmp(key, "max_lanes")) s->max_lanes = (int)value;
    else if (!strcmp(key, "kv_ctx_cap")) { if (s->kvcache) { cz_set_error("kv_ctx_cap must be set before the first step / decode"); return CZ_ERR_STATE; } s->kv_ctx_cap = (int)value; }
    else if (!strcmp(key, "probe_launch")) s->probe_launch = (int)value;
    else if (!strcmp(key, "stream_priority")) return set_stream_priority(s, (int)value);
    else if (!strcmp(key, "cu_mask_xcd")) return set_stream_xcd(s, (int)(value % 16), (int)(value / 16) + 1);   // value = first_xcd + 16 * (n_xcd - 1); -1: no mask
    else if (!strcmp(key, "step_sync")) s->step_sync = (int)value;
    else if (!strcmp(key, "wide_min_streams")) s->wide_min_streams = (int)value;
    else if (!strcmp(key, "wide_mode")) s->wide_mode = (int)value;
    else if (!strcmp(key, "gemm_mid_rows")) g_gemm_mid_rows = (int)value;
    else if (!strcmp(key, "no_last_prune")) s->no_last_prune = (int)value;
    else if (!strcmp(key, "rwkv_seq")) s->rwkv_seq = (int)value;
    else if (!strcmp(key, "rwkv_wide_min")) s->rwkv_wide_min = (int)value;
    else if (!strcmp(key, "rwkv_wide_mode")) s->rwkv_wide_mode = (int)value;
    else if (!strcmp(key, "rw_state_pad")) g_rw_state_pad = (int)value < 0 ? 0 : ((int)value > 48000 ? 48000 : (int)value);
    else if (!strcmp(key, "rwkv_seq_rows")) g_rw_seq_rows = (int)value < 1 ? 1 : (int)value;
    else if (!strcmp(key, "mid_nw")) g_mid_nw = (int)value;
    else if (!strcmp(key, "mid_colmaj")) g_mid_colmaj = (int)value;
    else if (!strcmp(key, "mid_quad")) g_mid_quad = (int)value;
    else if (!strcmp(key, "cdf_dec_grid_cap")) g_cdf_dec_grid_cap = (int)value;
    else if (!strcmp(key, "gemv_max_rows")) g_gemv_max_rows = (int)value;
    else if (!strcmp(key, "gemv_fold_attn")) g_gemv_fold_attn = (int)value;
    else if (!strcmp(key, "dec_attn_split_heads")) g_dec_attn_split_heads = (int)value;
    else if (!strcmp(key, "dec_attn_lds_pad")) g_dec_attn_lds_pad = (int)value;              // (before the first decode: the kernels' LDS limit is set once)
    else if (!strcmp(key, "dec_wide_rows")) g_dec_wide_rows = (int)value;
    else if (!strcmp(key, "dec_attn_small_grid")) g_dec_attn_small_grid = (int)value;
    else if (!strcmp(key, "dec_attn_split_max")) g_dec_attn_split_max = (int)value;
    else if (!strcmp(key, "dec_attn_waves")) g_dec_attn_waves = (int)value == 4 ? 4 : ((int)value == 8 ? 8 : 0);      // (process-wide) 0 = by the number of concurrent lanes
    else if (!strcmp(key, "gemm_pipe")) g_gemm_pipe = (int)value;        // (process-wide) 0: the 128 x 128 tiles of the big-M GEMM on the one-buffer K loop
    else if (!strcmp(key, "gemm_lds_kb")) g_gemm_lds_kb = (int)value < 0 ? 0 : ((int)value > 160 ? 160 : (int)value);   // (process-wide) fewer big-M GEMM blocks per CU
    else if (!strcmp(key, "dbg_skip")) s->dbg_skip = (int)value;          // tools/bench_decode_step.py: 1 = no attention, 2 = no GEMMs/norms (timing only; results are garbage)
    else if (!strcmp(key, "fuse_norm")) s->fuse_norm = (int)value;
    else if (!strcmp(key, "dec_norm_mfma")) g_dec_norm_mfma = (int)value;
    else if (!strcmp(key, "dec_grid_target")) g_dec_grid_target = (int)value;
    else if (!strcmp(key, "dec_mt_max")) g_dec_mt_max = (int)value >= 8 ? 8 : ((int)value >= 4 ? 4 : ((int)value >= 2 ? 2 : 1));
    else if (!strcmp(key, "dec_nw_cap")) g_dec_nw_cap = (int)value < 4 ? 4 : ((int)value > 12 ? 12 : (int)value);
    else { cz_set_error("unknown debug key %s", key); return CZ_ERR_ARG; }
    return CZ_OK;
}

// number of rows the exact-parallel CDF kernel handed to the serial walk since the session was opened
extern "C" long cz_debug_get(cz_session* s, const char* key) {
    if (!s || !key) return -1;
    for (const ProcKnob& k : proc_knobs()) if (!strcmp(key, k.key)) return (long)*k.p;      // a process-wide knob's current value
    if (!strcmp(key, "graph_launch_cpu_us")) { unsigned long long t = s->graph_launch_cpu_ns; s->graph_launch_cpu_ns = 0; return (long)(t / 1000); }
    if (!strcmp(key, "graph_launch_idle_us")) { unsigned long long t = s->graph_launch_idle_ns; s->graph_launch_idle_ns = 0; return (long)(t / 1000); }   // first 64 launches
    if (!strcmp(key, "graph_launch_us")) { unsigned long long t = s->graph_launch_ns; s->graph_launch_ns = 0; return (long)(t / 1000); }   // host time inside hipGraphLaunch
    if (!strcmp(key, "cdf_fallback_rows")) {
        unsigned v = 0;
        hipStreamSynchronize(s->stream);
        if (hipMemcpy(&v, s->fallback_dev, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        long total = (long)v;
        for (cz_session* l : s->lanes) { long t = cz_debug_get(l, key); if (t < 0) return -1; total += t; }
        return total;
    }
    return -1;
}

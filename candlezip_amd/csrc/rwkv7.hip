// rwkv7.hip — RWKV7 "Goose" recurrent-state path on gfx950: lock-step T == 1 steps of B independent streams.
//
// Replaces Rwkv7Session (src/models.rs:124-180) and candle_rwkv7 Model::forward for x[1,1]
// (candle_rwkv7/src/models/rwkv7.rs: Model 509-524, Block 468-478, SelfAttention t==1 179-328, FeedForward 412-430,
// group_norm 529-542) plus the RWKV branch of the token loops (src/main.rs:1824, 1916, 2302-2325, 2346-2349, 2812-2834).
// Numerics: canonical chunk-64 fmaf reductions, owned exp/log/tanh, candle's fused layer_norm / sigmoid forms — the parity tests check every
// result bit for bit against the CPU oracle.  RWKV has no re-prime: encode and decode both walk the recurrence token by token, so the
// throughput lever is the number of streams stepped together (rows of the skinny MFMA GEMMs, kernels_decode.hip).
//
// Per layer and step: prep (LayerNorm + token-shift mixes, 1 block/stream) -> 7 projections -> LoRA activations -> 4
// LoRA up-projections -> state kernel (1 wave per (stream, head): lane = state row i, the 64-float row lives in registers;
// S' = S*w - (S.kk)(kk*a)^T + v k^T ; y = S'.r ; GroupNorm ; bonus ; gate) -> output projection (+residual) -> channel-mix
// prep -> key projection with relu^2 epilogue -> value projection (+residual).
// State per stream: 12 x (768 + 12*64*64 + 768) f32 = 2.36 MB, resident in HBM; HBM-bound at small B.
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "cz_host.h"
#include "cz_session.h"

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            cz_set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            return CZ_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)
#define CZCHK(expr) do { cz_status _s = (expr); if (_s != CZ_OK) return _s; } while (0)

#ifndef RW_STATE_NT
#define RW_STATE_NT 1          // rw_state_kernel streams its state tile with non-temporal loads / stores (tuning build: -DRW_STATE_NT=0)
#endif
int g_rw_state_pad = 0;        // tuning knob (cz_debug_set "rw_state_pad"): extra dynamic LDS bytes per rw_state_kernel block (fewer waves per CU -> several generations of waves, whose load / compute / store phases then overlap)
__host__ __device__ __forceinline__ int kq_idx(int k) { return (k & ~63) | ((k & 3) << 4) | ((k & 63) >> 2); }

struct RwLayer {
    // matrices, kq-packed bf16 [N][K]
    uint16_t *r, *k, *v, *o, *w1, *w2, *a1, *a2, *v1, *v2, *g1, *g2, *fk, *fv;
    // the same matrices in natural k order, [pad64(N)][pad64(K)] zero-padded: operands of the tiled big-M GEMM (sequence-parallel encode)
    uint16_t *r_n, *k_n, *v_n, *o_n, *w1_n, *w2_n, *a1_n, *a2_n, *v1_n, *v2_n, *g1_n, *g2_n, *fk_n, *fv_n;
    // ... and packed for the K-group GEMM of wide decode groups (kernels_gemm_mid.hip, "k2" order), same padded shapes
    uint16_t *r_2, *k_2, *v_2, *o_2, *w1_2, *w2_2, *a1_2, *a2_2, *v1_2, *v2_2, *g1_2, *g2_2, *fk_2, *fv_2;
    // vectors (bf16)
    uint16_t *pre_w, *pre_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b, *gn_w, *gn_b, *x_r, *x_w, *x_k, *x_v, *x_a, *x_g, *w0, *a0, *v0, *k_k, *k_a,
        *r_k, *fx_k;
    int has_v;
};
struct Rwkv7Model {
    cz_rwkv7_config cfg{};
    std::vector<void*> allocs;
    uint16_t *emb = nullptr, *head = nullptr, *head_n = nullptr, *lno_w = nullptr, *lno_b = nullptr;
    std::vector<RwLayer> L;
    // per-stream state [L][B][...]
    int B = 0;
    float *att_x_prev = nullptr, *att_state = nullptr, *ffn_x_prev = nullptr, *v_first = nullptr;
    // work buffers
    float *x = nullptr, *mix = nullptr, *proj = nullptr, *lora1 = nullptr, *lora1a = nullptr, *lora2 = nullptr, *att = nullptr, *ffk = nullptr, *xn = nullptr;
    float* logits = nullptr;
    uint32_t *tok = nullptr, *sym = nullptr; unsigned char* mask = nullptr;
    size_t index_pos = 0;
    // lock-step loops under hipGraph replay: one executable graph per loop kind, per-iteration scalars in HBM (step_dev)
    struct Graph { hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; int B = 0; unsigned long long gen = ~0ull; } genc, gdec;
    unsigned long long* step_dev = nullptr;
    // sequence-parallel encode (rw_encode_seq): packed-row work buffers, grown on demand
    DevBuf q_x, q_xa, q_mix, q_p, q_tl, q_l2, q_sc, q_vf, q_y, q_ffk, q_idx, q_xn;
    std::vector<DevBuf*> seq_bufs() { return {&q_x, &q_xa, &q_mix, &q_p, &q_tl, &q_l2, &q_sc, &q_vf, &q_y, &q_ffk, &q_idx, &q_xn}; }
};

// ------------------------------------------------------------------ kernels
__global__ void rw_synth_kernel(uint16_t* dst, size_t numel, uint64_t seed, uint32_t tid, float stdv, float mean) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = cz_synth_bf16(seed, tid, i, stdv, mean);
}
// Wd[n][kq(k)] = W[n][k] for k < K, 0 for K <= k < Kp (K padded to a multiple of 64: fma(0, 0, acc) == acc)
__global__ void rw_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wd, size_t N, int K, int Kp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N * (size_t)Kp; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / Kp; const int k = (int)(i % Kp);
        Wd[n * Kp + kq_idx(k)] = k < K ? W[n * K + k] : (uint16_t)0;
    }
}

// Wn[n][k] = W[n][k] for n < N, k < K, else 0  ([Np][Kp], natural k order)
__global__ void rw_pad_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wn, size_t N, int K, size_t Np, int Kp) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < Np * (size_t)Kp; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / Kp; const int k = (int)(i % Kp);
        Wn[i] = (n < N && k < K) ? W[n * K + k] : (uint16_t)0;
    }
}

// candle_nn::ops::layer_norm over C = blockDim.x elements held one per thread (canonical chunk sums; var = E[x^2] - mean^2)
__device__ float block_layer_norm(float x, const uint16_t* w, const uint16_t* b, float eps, float* sh /* C + 64 floats */) {
    const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6, C = blockDim.x, nch = C >> 6;
    float* xs = sh; float* p1 = sh + C; float* p2 = p1 + 16; float* st = p2 + 16;
    xs[tid] = x;
    __syncthreads();
    if (lane == 0) {
        float a1 = 0.0f, a2 = 0.0f;
        for (int k = 0; k < 64; ++k) { const float v = xs[c * 64 + k]; const float q = v * v; a1 = a1 + v; a2 = a2 + q; }
        p1[c] = a1; p2[c] = a2;
    }
    __syncthreads();
    if (tid == 0) {
        float s1 = p1[0], s2 = p2[0];
        for (int k = 1; k < nch; ++k) { s1 = s1 + p1[k]; s2 = s2 + p2[k]; }
        const float mean = s1 / (float)C;
        const float var = s2 / (float)C - mean * mean;
        st[0] = mean; st[1] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    const float y = (x - st[0]) * st[1] * cz_bf16_to_f32(w[tid]) + cz_bf16_to_f32(b[tid]);
    __syncthreads();
    return y;
}

// time-mix prep (rwkv7.rs:468-471, 193-203): [pre-LN for layer 0,] xa = LN1(x); xx = prev - xa; six token-shift mixes in kq layout
__global__ void rw_prep_att_kernel(float* __restrict__ X, const RwLayer L, int layer0, float eps, float* __restrict__ x_prev /*[B][C]*/,
                                   float* __restrict__ MIX /*[6][B][C] kq (natural when !out_kq)*/, int B, const unsigned char* __restrict__ mask, int out_kq) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    float x = X[(size_t)b * C + tid];
    if (layer0) { x = block_layer_norm(x, L.pre_w, L.pre_b, eps, sh); X[(size_t)b * C + tid] = x; }
    const float xa = block_layer_norm(x, L.ln1_w, L.ln1_b, eps, sh);
    const float xx = x_prev[(size_t)b * C + tid] - xa;
    const uint16_t* mixw[6] = {L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g};
    const int kq = out_kq ? kq_idx(tid) : tid;
#pragma unroll
    for (int j = 0; j < 6; ++j) MIX[((size_t)j * B + b) * C + kq] = xa + xx * cz_bf16_to_f32(mixw[j][tid]);
    if (!mask || mask[b]) x_prev[(size_t)b * C + tid] = xa;                  // state.att_x_prev (:308)
}
// channel-mix prep (rwkv7.rs:472-474, 414-425)
__global__ void rw_prep_ffn_kernel(const float* __restrict__ X, const RwLayer L, float eps, float* __restrict__ x_prev, float* __restrict__ XK /*[B][C] kq*/,
                                   const unsigned char* __restrict__ mask, int out_kq) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const float x = X[(size_t)b * C + tid];
    const float xf = block_layer_norm(x, L.ln2_w, L.ln2_b, eps, sh);
    const float xx = x_prev[(size_t)b * C + tid] - xf;
    XK[(size_t)b * C + (out_kq ? kq_idx(tid) : tid)] = xf + xx * cz_bf16_to_f32(L.fx_k[tid]);
    if (!mask || mask[b]) x_prev[(size_t)b * C + tid] = xf;
}
// final LayerNorm -> kq layout (rwkv7.rs:515)
__global__ void rw_ln_out_kernel(const float* __restrict__ X, const uint16_t* w, const uint16_t* bb, float eps, float* __restrict__ Y, int out_kq) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const float y = block_layer_norm(X[(size_t)b * C + tid], w, bb, eps, sh);
    Y[(size_t)b * C + (out_kq ? kq_idx(tid) : tid)] = y;
}
// LoRA hidden activations (rwkv7.rs:211, 229, 235, 244): tanh(w1.), a1., sigmoid(g1.), v1.  -> kq layouts padded to 64
__global__ void rw_lora_act_kernel(const float* __restrict__ P /*[B][ldp]*/, int ldp, int ow, int oa, int og, int ov, int rw, int ra, int rg, int rv,
                                   float* __restrict__ TW, float* __restrict__ TA, float* __restrict__ TG, float* __restrict__ TV, int pw, int pa, int pg, int pv, int out_kq) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = P + (size_t)b * ldp;
    const int o = out_kq ? kq_idx(t) : t;
    if (t < pw) TW[(size_t)b * pw + o] = t < rw ? cz_tanhf(p[ow + t]) : 0.0f;
    if (t < pa) TA[(size_t)b * pa + o] = t < ra ? p[oa + t] : 0.0f;
    if (t < pg) TG[(size_t)b * pg + o] = t < rg ? cz_sigmoidf(p[og + t]) : 0.0f;
    if (t < pv) TV[(size_t)b * pv + o] = t < rv ? p[ov + t] : 0.0f;
}

// state kernel: block = (stream, head), 64 lanes = state rows (rwkv7.rs:213-222, 231, 239-266, 289-323, 529-542)
// Register discipline: the 64-float state row is the only large live set.  Everything a lane needs of the OTHER lanes goes through LDS as
// a value each lane computed from its own element (kk^2, kk*a, r*k*r_k ...: the same products, rounded the same way, as when every lane
// recomputed all 64 of them), the 64-long canonical sums then read one LDS array each, and barriers separate the stages — so that the
// scheduler cannot hoist hundreds of LDS / global loads above the chains (it did: 256 VGPRs + scratch, one wave per SIMD, 3 TB/s).
__global__ __launch_bounds__(64, 4) void rw_state_kernel(const float* __restrict__ P /*[B][ldp]: r | k | v*/, int ldp, const float* __restrict__ L2 /*[B][4][C]: w a g nu*/,
                                                      const RwLayer L, int layer, int C, float* __restrict__ S /*[B][H] tiles*/,
                                                      float* __restrict__ v_first /*[B][C]*/, float* __restrict__ OUT /*[B][C] kq*/,
                                                      const unsigned char* __restrict__ mask, int B, int out_kq) {
    __shared__ __attribute__((aligned(16))) float bc[5][64];     // kk | kk*a | w | k | r   (as rw_seq_scan_kernel)
    __shared__ __attribute__((aligned(16))) float tmp[64];
    const int b = blockIdx.x, h = blockIdx.y, i = threadIdx.x, c = h * 64 + i;
    // A stream whose symbol was a literal does not step the model (src/main.rs:2347, :2832): nothing of its step is kept — state, token shift,
    // v_first and logits are all masked — so its block leaves at once (block-uniform).  The stores below are therefore unconditional: with them
    // inside a `mask` branch the compiler kept the old AND the new row alive (476 B of scratch per lane at 128 VGPRs).
    if (mask && !mask[b]) return;
    // state tile of a head in HBM: element (i, j) at ((j >> 2) * 64 + i) * 4 + (j & 3) — lane i's sixteen 16-B pieces of ITS row sit 1 KB apart,
    // so load / store j4 moves one contiguous KB per wave (row-major rows of 256 B would be 64 lanes in 64 different rows: ~17x slower, guide).
    // Requested first: the round trip overlaps the per-token terms below.
    float* Srow = S + ((size_t)b * gridDim.y + h) * 4096 + (size_t)i * 4;
    float srow[64];
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
        typedef float rw_f4 __attribute__((ext_vector_type(4)));
        const rw_f4 t4 = RW_STATE_NT ? __builtin_nontemporal_load(reinterpret_cast<const rw_f4*>(Srow + 256 * j4)) : *reinterpret_cast<const rw_f4*>(Srow + 256 * j4);   // read once, written once per step: streamed past the caches
        srow[4 * j4] = t4.x; srow[4 * j4 + 1] = t4.y; srow[4 * j4 + 2] = t4.z; srow[4 * j4 + 3] = t4.w;
    }
    const float* p = P + (size_t)b * ldp;
    const float* l2 = L2 + (size_t)b * 4 * C;
    const float r = p[c], kraw = p[C + c];
    float v = p[2 * C + c];
    // w = exp(-exp(-softplus(-(w0 + lora)) - 0.5))
    const float wc = l2[c] + cz_bf16_to_f32(L.w0[c]);
    const float sp = cz_logf(cz_expf(-wc) + 1.0f);
    const float wt = (-sp) + (-0.5f);
    const float w = cz_expf(-cz_expf(wt));
    const float a = cz_sigmoidf(l2[C + c] + cz_bf16_to_f32(L.a0[c]));
    const float g = l2[2 * C + c];
    if (layer == 0) v_first[(size_t)b * C + c] = v;
    else if (L.has_v) {
        const float nu = cz_sigmoidf(l2[3 * C + c] + cz_bf16_to_f32(L.v0[c]));
        const float dl = v_first[(size_t)b * C + c] - v;
        v = v + dl * nu;
    }
    const float kk0 = kraw * cz_bf16_to_f32(L.k_k[c]);
    tmp[i] = kk0 * kk0;
    __syncthreads();
    float ss = 0.0f;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) ss = ss + tmp[j];
    float nrm = sqrtf(ss);
    if (!(nrm > 1e-12f)) nrm = 1e-12f;
    const float kkn = kk0 / nrm;
    const float sc = 1.0f + (a - 1.0f) * cz_bf16_to_f32(L.k_a[c]);
    const float k = kraw * sc;
    __syncthreads();
    bc[0][i] = kkn; bc[1][i] = kkn * a; bc[2][i] = w; bc[3][i] = k; bc[4][i] = r;
    tmp[i] = r * k * cz_bf16_to_f32(L.r_k[c]);                       // the bonus term's r . (k r_k), element i
    __syncthreads();
    float al = 0.0f;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) al = al + tmp[j];
    float t = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; ++j) t = cz_fma(srow[j], bc[0][j], t);                          // (S @ kk)[i]
    float y = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float sn = (srow[j] * bc[2][j] - t * bc[1][j]) + v * bc[3][j];
        srow[j] = sn;
        y = cz_fma(sn, bc[4][j], y);                                                        // (S' @ r)[i]
    }
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
        typedef float rw_f4 __attribute__((ext_vector_type(4)));
        const rw_f4 o4 = rw_f4{srow[4 * j4], srow[4 * j4 + 1], srow[4 * j4 + 2], srow[4 * j4 + 3]};
        if (RW_STATE_NT) __builtin_nontemporal_store(o4, reinterpret_cast<rw_f4*>(Srow + 256 * j4)); else *reinterpret_cast<rw_f4*>(Srow + 256 * j4) = o4;
    }
    __syncthreads();
    tmp[i] = y;
    __syncthreads();
    float s1 = 0.0f;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) s1 = s1 + tmp[j];
    const float mean = s1 * 0.015625f;
    const float cen = y - mean;
    __syncthreads();
    tmp[i] = cen * cen;
    __syncthreads();
    float s2 = 0.0f;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) s2 = s2 + tmp[j];
    const float den = sqrtf(s2 * 0.015625f + (float)64e-5);
    const float gn = cen / den * cz_bf16_to_f32(L.gn_w[c]) + cz_bf16_to_f32(L.gn_b[c]);
    const float ad = al * v;
    OUT[(size_t)b * C + (out_kq ? kq_idx(c) : c)] = (gn + ad) * g;
}

__global__ void rw_pick_kernel(const uint32_t* __restrict__ ids, const unsigned long long* __restrict__ off, const unsigned long long* __restrict__ n,
                               unsigned long long i, uint32_t V, uint32_t* __restrict__ tok, uint32_t* __restrict__ sym, unsigned char* __restrict__ mask, int B,
                               const unsigned long long* __restrict__ step_dev) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (step_dev) i = *step_dev;
    const unsigned long long nb = n[b];
    uint32_t t = 0, s = 0; unsigned char m = 0;
    if (i + 1 < nb) { t = ids[off[b] + i]; s = ids[off[b] + i + 1]; m = (i == 0 || t < V) ? 1 : 0; }     // literal symbols do not step the model (:2347)
    tok[b] = t < V ? t : 0; sym[b] = s; mask[b] = m;
}
__global__ void rw_mask_from_sym_kernel(const cz_dec_state* __restrict__ st, const uint32_t* __restrict__ sym, uint32_t V, uint32_t* __restrict__ tok,
                                        unsigned char* __restrict__ mask, int B, unsigned long long i, const unsigned long long* __restrict__ step_dev) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (step_dev) i = *step_dev;
    const bool act = st[b].active && i < st[b].token_count;
    const uint32_t s = sym[b];
    mask[b] = (act && s < V) ? 1 : 0;                                                                       // :2832
    tok[b] = s < V ? s : 0;
}
__global__ void rw_counter_kernel(unsigned long long* c, int inc) { if (inc) *c += 1ull; else *c = 0ull; }
__global__ void rw_fill_kernel(uint32_t* tok, unsigned char* mask, uint32_t t, int B) { int b = blockIdx.x * blockDim.x + threadIdx.x; if (b < B) { tok[b] = t; mask[b] = 1; } }
__global__ void rw_init_out_kernel(cz_dec_state* st, int n, uint32_t bos) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) st[i].out_ids[0] = bos; }

// ------------------------------------------------------------------ model construction
static cz_status rw_alloc(Rwkv7Model* m, void** p, size_t bytes) {
    HIPCHK(hipMalloc(p, bytes ? bytes : 16));
    m->allocs.push_back(*p);
    return CZ_OK;
}
static int pad64(int k) { return (k + 63) / 64 * 64; }

cz_status cz_rwkv7_build(cz_session* s, const cz_model_desc* desc) {
    Rwkv7Model* m = new Rwkv7Model();
    s->rwkv = m;
    if (desc->config_path) {                                     // candle_rwkv7 Config (rwkv7.rs:7-32; defaults :27-32, ffn :406)
        FILE* f = fopen(desc->config_path, "rb");
        if (!f) { cz_set_error("cannot open %s", desc->config_path); return CZ_ERR_IO; }
        std::string txt; char buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, n);
        fclose(f);
        JsonValue root; std::string err;
        if (!cz_json_parse(txt.data(), txt.size(), &root, &err) || root.kind != JsonValue::Obj) { cz_set_error("failed to parse %s as an RWKV7 config: %s", desc->config_path, err.c_str()); return CZ_ERR_IO; }
        auto num = [&](const char* k, double def) { const JsonValue* v = root.get(k); return v && v->kind == JsonValue::Num ? v->num : def; };
        cz_rwkv7_config& c = m->cfg;
        c.vocab = (int)num("vocab_size", 0); c.hidden = (int)num("hidden_size", 0); c.n_layers = (int)num("num_hidden_layers", 0);
        c.head_dim = (int)num("head_dim", num("head_size", 0));
        c.d_ffn = (int)num("intermediate_size", 4.0 * c.hidden);
        c.a_rank = (int)num("a_low_rank_dim", 64); c.decay_rank = (int)num("decay_low_rank_dim", 64);
        c.v_rank = (int)num("v_low_rank_dim", 32); c.gate_rank = (int)num("gate_low_rank_dim", 128);
        c.norm_eps = (float)num("norm_eps", 1e-5);
        if (c.vocab <= 0 || c.hidden <= 0 || c.n_layers <= 0 || c.head_dim <= 0) { cz_set_error("%s: vocab_size / hidden_size / num_hidden_layers / head_dim are required", desc->config_path); return CZ_ERR_IO; }
    } else m->cfg = desc->rwkv;
    const cz_rwkv7_config& c = m->cfg;
    if (c.head_dim != 64 || c.hidden % 64 || c.hidden > 1024 || c.vocab % 32 || c.d_ffn % 64 || c.n_layers <= 0 || c.a_rank > 64 || c.decay_rank > 64 ||
        c.v_rank > 64 || c.gate_rank > 128 || c.a_rank % 32 || c.decay_rank % 32 || c.v_rank % 32 || c.gate_rank % 32) {
        cz_set_error("unsupported RWKV7 shape (need head_dim 64, hidden <= 1024 and a multiple of 64, lora ranks multiples of 32)");
        return CZ_ERR_UNSUPPORTED;
    }
    std::vector<std::unique_ptr<SafeTensorsFile>> files;
    const bool synthetic = desc->n_weight_paths == 0;
    for (size_t i = 0; i < desc->n_weight_paths; ++i) {
        files.emplace_back(new SafeTensorsFile());
        if (!cz_safetensors_open(desc->weight_paths[i], files.back().get())) return CZ_ERR_IO;
    }
    const int C = c.hidden, F = c.d_ffn, V = c.vocab;
    uint16_t* tmp = nullptr;
    size_t tmp_cap = std::max((size_t)V * C, (size_t)F * C);
    CZCHK(rw_alloc(m, (void**)&tmp, tmp_cap * 2));
    // fetch a tensor in natural [N][K] layout into `dst` (bf16)
    auto fetch = [&](const std::string& name, uint16_t* dst, size_t numel, float stdv, float mean) -> cz_status {
        if (synthetic) {
            hipLaunchKernelGGL(rw_synth_kernel, dim3(1024), dim3(256), 0, s->stream, dst, numel, s->seed, cz_fnv1a32(name.c_str()), stdv, mean);
            HIPCHK(hipGetLastError());
            return CZ_OK;
        }
        for (auto& f : files) {
            auto it = f->tensors.find(name);
            if (it == f->tensors.end()) continue;
            const SafeTensor& t = it->second;
            size_t n = 1; for (auto dd : t.shape) n *= (size_t)dd;
            if (n != numel) { cz_set_error("tensor %s has %zu elements, expected %zu", name.c_str(), n, numel); return CZ_ERR_IO; }
            if (t.dtype == "BF16") { HIPCHK(hipMemcpy(dst, t.data, numel * 2, hipMemcpyHostToDevice)); return CZ_OK; }
            if (t.dtype == "F32") {
                std::vector<uint16_t> h16(numel);
                const uint32_t* src = (const uint32_t*)t.data;
                for (size_t i = 0; i < numel; ++i) {
                    uint32_t u; memcpy(&u, src + i, 4);
                    if ((u & 0xffffu) && !s->round_f32) { cz_set_error("tensor %s is F32 with values that are not bf16-representable (unsupported; see cz_model_desc.round_f32_weights_to_bf16)", name.c_str()); return CZ_ERR_UNSUPPORTED; }
                    if (u & 0xffffu) u = cz_f32_bits_round_bf16(u);
                    h16[i] = (uint16_t)(u >> 16);
                }
                HIPCHK(hipMemcpy(dst, h16.data(), numel * 2, hipMemcpyHostToDevice));
                return CZ_OK;
            }
            cz_set_error("tensor %s has unsupported dtype %s", name.c_str(), t.dtype.c_str());
            return CZ_ERR_UNSUPPORTED;
        }
        cz_set_error("tensor %s not found in the safetensors files", name.c_str());
        return CZ_ERR_IO;
    };
    auto mat = [&](const std::string& name, uint16_t** out, uint16_t** out_nat, uint16_t** out_k2, int N, int K, float stdv) -> cz_status {
        const int Kp = pad64(K), Np = pad64(N);
        CZCHK(rw_alloc(m, (void**)out, (size_t)N * Kp * 2));
        CZCHK(rw_alloc(m, (void**)out_nat, (size_t)Np * Kp * 2));
        CZCHK(fetch(name, tmp, (size_t)N * K, stdv, 0.0f));
        hipLaunchKernelGGL(rw_pack_kernel, dim3(1024), dim3(256), 0, s->stream, tmp, *out, (size_t)N, K, Kp);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(rw_pad_kernel, dim3(1024), dim3(256), 0, s->stream, tmp, *out_nat, (size_t)N, K, (size_t)Np, Kp);
        HIPCHK(hipGetLastError());
        if (out_k2) {
            CZCHK(rw_alloc(m, (void**)out_k2, (size_t)Np * Kp * 2));
            HIPCHK(cz_launch_k2_pack(s->stream, *out_nat, *out_k2, (size_t)Np, Kp, 0));
        }
        // `tmp` is reused by the next matrix.  Synthetic tensors are generated on s->stream (ordered behind this kernel), but a checkpoint
        // tensor arrives by a blocking hipMemcpy on the null stream, which a non-blocking stream does not wait for: drain the pack first
        if (!synthetic) HIPCHK(hipStreamSynchronize(s->stream));
        return CZ_OK;
    };
    auto vec = [&](const std::string& name, uint16_t** out, int n, float stdv, float mean) -> cz_status {
        CZCHK(rw_alloc(m, (void**)out, (size_t)n * 2));
        return fetch(name, *out, (size_t)n, stdv, mean);
    };
    CZCHK(rw_alloc(m, (void**)&m->emb, (size_t)V * C * 2));
    CZCHK(fetch("model.embeddings.weight", m->emb, (size_t)V * C, desc->synthetic_embed_std, 0.0f));
    m->L.resize(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        RwLayer& l = m->L[i];
        memset(&l, 0, sizeof l);
        const std::string p = "model.layers." + std::to_string(i) + ".";
        if (i == 0) { CZCHK(vec(p + "pre_norm.weight", &l.pre_w, C, 0.1f, 1.0f)); CZCHK(vec(p + "pre_norm.bias", &l.pre_b, C, 0.02f, 0.0f)); }
        CZCHK(vec(p + "attn_norm.weight", &l.ln1_w, C, 0.1f, 1.0f)); CZCHK(vec(p + "attn_norm.bias", &l.ln1_b, C, 0.02f, 0.0f));
        CZCHK(vec(p + "ffn_norm.weight", &l.ln2_w, C, 0.1f, 1.0f));  CZCHK(vec(p + "ffn_norm.bias", &l.ln2_b, C, 0.02f, 0.0f));
        CZCHK(mat(p + "attn.r_proj.weight", &l.r, &l.r_n, &l.r_2, C, C, 0.03f)); CZCHK(mat(p + "attn.k_proj.weight", &l.k, &l.k_n, &l.k_2, C, C, 0.03f));
        CZCHK(mat(p + "attn.v_proj.weight", &l.v, &l.v_n, &l.v_2, C, C, 0.03f)); CZCHK(mat(p + "attn.o_proj.weight", &l.o, &l.o_n, &l.o_2, C, C, 0.03f));
        CZCHK(vec(p + "attn.g_norm.weight", &l.gn_w, C, 0.1f, 1.0f)); CZCHK(vec(p + "attn.g_norm.bias", &l.gn_b, C, 0.02f, 0.0f));
        CZCHK(vec(p + "attn.x_r", &l.x_r, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_w", &l.x_w, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_k", &l.x_k, C, 0.15f, 0.5f));
        CZCHK(vec(p + "attn.x_v", &l.x_v, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_a", &l.x_a, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_g", &l.x_g, C, 0.15f, 0.5f));
        CZCHK(mat(p + "attn.w_lora.lora.0.weight", &l.w1, &l.w1_n, &l.w1_2, c.decay_rank, C, 0.05f)); CZCHK(mat(p + "attn.w_lora.lora.2.weight", &l.w2, &l.w2_n, &l.w2_2, C, c.decay_rank, 0.05f));
        CZCHK(vec(p + "attn.w_lora.lora.2.bias", &l.w0, C, 1.0f, 0.0f));
        CZCHK(mat(p + "attn.a_lora.lora.0.weight", &l.a1, &l.a1_n, &l.a1_2, c.a_rank, C, 0.05f)); CZCHK(mat(p + "attn.a_lora.lora.2.weight", &l.a2, &l.a2_n, &l.a2_2, C, c.a_rank, 0.05f));
        CZCHK(vec(p + "attn.a_lora.lora.2.bias", &l.a0, C, 0.5f, 0.0f));
        if (i > 0) {                                                    // rwkv7.rs:146-152 (`contains_tensor`): the synthetic model mirrors the oracle
            l.has_v = 1;
            CZCHK(mat(p + "attn.v_lora.lora.0.weight", &l.v1, &l.v1_n, &l.v1_2, c.v_rank, C, 0.05f)); CZCHK(mat(p + "attn.v_lora.lora.2.weight", &l.v2, &l.v2_n, &l.v2_2, C, c.v_rank, 0.05f));
            CZCHK(vec(p + "attn.v_lora.lora.2.bias", &l.v0, C, 0.5f, 1.0f));
        }
        CZCHK(mat(p + "attn.g_lora.lora.0.weight", &l.g1, &l.g1_n, &l.g1_2, c.gate_rank, C, 0.05f)); CZCHK(mat(p + "attn.g_lora.lora.2.weight", &l.g2, &l.g2_n, &l.g2_2, C, c.gate_rank, 0.05f));
        CZCHK(vec(p + "attn.k_k", &l.k_k, C, 0.1f, 0.85f)); CZCHK(vec(p + "attn.k_a", &l.k_a, C, 0.1f, 1.0f)); CZCHK(vec(p + "attn.r_k", &l.r_k, C, 0.05f, 0.0f));
        CZCHK(vec(p + "ffn.x_k", &l.fx_k, C, 0.15f, 0.5f));
        CZCHK(mat(p + "ffn.key.weight", &l.fk, &l.fk_n, &l.fk_2, F, C, 0.03f)); CZCHK(mat(p + "ffn.value.weight", &l.fv, &l.fv_n, &l.fv_2, C, F, 0.03f));
    }
    CZCHK(vec("model.norm.weight", &m->lno_w, C, 0.1f, 1.0f)); CZCHK(vec("model.norm.bias", &m->lno_b, C, 0.02f, 0.0f));
    CZCHK(mat("lm_head.weight", &m->head, &m->head_n, nullptr, V, C, desc->synthetic_embed_std));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CZ_OK;
}
static void rw_graph_destroy(Rwkv7Model::Graph& G) {
    if (G.e) { hipGraphExecDestroy(G.e); G.e = nullptr; }
    if (G.g) { hipGraphDestroy(G.g); G.g = nullptr; }
    G.B = 0; G.gen = ~0ull;
}
void cz_rwkv7_destroy(cz_session* s) {
    if (!s->rwkv) return;
    rw_graph_destroy(s->rwkv->genc); rw_graph_destroy(s->rwkv->gdec);
    for (DevBuf* b : s->rwkv->seq_bufs()) b->release();
    for (auto p : s->rwkv->allocs) { hipError_t e = hipFree(p); (void)e; }
    delete s->rwkv; s->rwkv = nullptr;
}
size_t cz_rwkv7_vocab(const cz_session* s) { return s->rwkv ? (size_t)s->rwkv->cfg.vocab : 0; }

static cz_status rw_ensure_streams(cz_session* s, int B) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    if (B <= m->B) return CZ_OK;
    if (B > s->max_streams) { cz_set_error("%d streams exceed max_streams = %d", B, s->max_streams); return CZ_ERR_ARG; }
    const int Bc = s->max_streams, C = c.hidden, L = c.n_layers, H = C / 64;
    const size_t ldp = 3 * (size_t)C + pad64(c.decay_rank) + pad64(c.a_rank) + 128 + pad64(c.v_rank);
    CZCHK(rw_alloc(m, (void**)&m->att_x_prev, (size_t)L * Bc * C * 4)); CZCHK(rw_alloc(m, (void**)&m->ffn_x_prev, (size_t)L * Bc * C * 4));
    CZCHK(rw_alloc(m, (void**)&m->att_state, (size_t)L * Bc * H * 64 * 64 * 4)); CZCHK(rw_alloc(m, (void**)&m->v_first, (size_t)Bc * C * 4));
    CZCHK(rw_alloc(m, (void**)&m->x, (size_t)Bc * C * 4)); CZCHK(rw_alloc(m, (void**)&m->mix, (size_t)6 * Bc * C * 4));
    CZCHK(rw_alloc(m, (void**)&m->proj, (size_t)Bc * ldp * 4)); CZCHK(rw_alloc(m, (void**)&m->lora1a, (size_t)Bc * (64 + 64 + 128 + 64) * 4));
    CZCHK(rw_alloc(m, (void**)&m->lora2, (size_t)Bc * 4 * C * 4)); CZCHK(rw_alloc(m, (void**)&m->att, (size_t)Bc * C * 4));
    CZCHK(rw_alloc(m, (void**)&m->ffk, (size_t)Bc * c.d_ffn * 4)); CZCHK(rw_alloc(m, (void**)&m->xn, (size_t)Bc * C * 4));
    CZCHK(rw_alloc(m, (void**)&m->logits, (size_t)Bc * c.vocab * 4));
    CZCHK(rw_alloc(m, (void**)&m->step_dev, 16));
    CZCHK(rw_alloc(m, (void**)&m->tok, (size_t)Bc * 4)); CZCHK(rw_alloc(m, (void**)&m->sym, (size_t)Bc * 4)); CZCHK(rw_alloc(m, (void**)&m->mask, (size_t)Bc));
    m->B = Bc;
    return CZ_OK;
}
static cz_status rw_reset(cz_session* s) {                        // State::new (rwkv7.rs:48-61)
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const int C = c.hidden, L = c.n_layers, H = C / 64;
    HIPCHK(hipMemsetAsync(m->att_x_prev, 0, (size_t)L * m->B * C * 4, s->stream));
    HIPCHK(hipMemsetAsync(m->ffn_x_prev, 0, (size_t)L * m->B * C * 4, s->stream));
    HIPCHK(hipMemsetAsync(m->att_state, 0, (size_t)L * m->B * H * 64 * 64 * 4, s->stream));
    return CZ_OK;
}

// one token per stream (tokens in m->tok, step mask in m->mask); logits of unmasked rows -> m->logits
static cz_status rw_step_narrow(cz_session* s, int B) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const int C = c.hidden, F = c.d_ffn, H = C / 64, Bc = m->B;
    const int pw = pad64(c.decay_rank), pa = pad64(c.a_rank), pg = 128, pv = pad64(c.v_rank);
    const int ow = 3 * C, oa = ow + pw, og = oa + pa, ov = og + pg, ldp = ov + pv;
    hipStream_t st = s->stream;
    const size_t shm = (size_t)(C + 64) * sizeof(float);
    float *TW = m->lora1a, *TA = TW + (size_t)Bc * 64, *TG = TA + (size_t)Bc * 64, *TV = TG + (size_t)Bc * 128;
    // per-family events (eager runs with profiling on): prep = LayerNorm / token shift / LoRA activations, gemm = every projection, state = rw_state_kernel
    const double st_bytes = (double)B * H * 2.0 * 64 * 64 * 4;                            // the state tile read and written once per (stream, head)
    s->prof_begin(PK_RW_PREP);
    HIPCHK(cz_launch_embed(st, m->tok, m->emb, m->x, B, C));
    for (int l = 0; l < c.n_layers; ++l) {
        const RwLayer& y = m->L[l];
        float* axp = m->att_x_prev + (size_t)l * Bc * C; float* fxp = m->ffn_x_prev + (size_t)l * Bc * C;
        float* S = m->att_state + (size_t)l * Bc * H * 64 * 64;
        if (l) s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_prep_att_kernel, dim3(B), dim3(C), shm, st, m->x, y, l == 0 ? 1 : 0, c.norm_eps, axp, m->mix, Bc, m->mask, 1);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * C * 4 * 9);
        s->prof_begin(PK_RW_GEMM);
        const float* MX = m->mix;                                   // [6][Bc][C]: r w k v a g
        auto mixp = [&](int j) { return MX + (size_t)j * Bc * C; };
        {   // seven projections of the six mixed inputs in one launch: r, k, v and the LoRA down-projections w1, a1, g1 (, v1)
            const float* A[7] = {mixp(0), mixp(2), mixp(3), mixp(1), mixp(4), mixp(5), mixp(3)};
            const int lda[7] = {C, C, C, C, C, C, C};
            const uint16_t* W[7] = {y.r, y.k, y.v, y.w1, y.a1, y.g1, y.v1};
            float* Cc[7] = {m->proj + 0, m->proj + C, m->proj + 2 * C, m->proj + ow, m->proj + oa, m->proj + og, m->proj + ov};
            const int ldc[7] = {ldp, ldp, ldp, ldp, ldp, ldp, ldp};
            const int N[7] = {C, C, C, c.decay_rank, c.a_rank, c.gate_rank, c.v_rank};
            const int K[7] = {C, C, C, C, C, C, C};
            HIPCHK(cz_launch_dec_gemm_group(st, y.has_v ? 7 : 6, A, lda, W, Cc, ldc, N, K, B));
        }
        s->prof_end(2.0 * B * (double)C * (3 * C + c.decay_rank + c.a_rank + c.gate_rank + (y.has_v ? c.v_rank : 0)), 2.0 * C * (3 * C + pw + pa + pg + pv));
        s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_lora_act_kernel, dim3(B), dim3(128), 0, st, m->proj, ldp, ow, oa, og, ov, c.decay_rank, c.a_rank, c.gate_rank, y.has_v ? c.v_rank : 0,
                           TW, TA, TG, TV, pw, pa, pg, pv, 1);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * (pw + pa + pg + pv) * 8);
        s->prof_begin(PK_RW_GEMM);
        float* L2 = m->lora2;                                        // [Bc][4][C]: w_lora, a_lin, g, nu_lin
        {   // the LoRA up-projections in one launch
            const float* A[4] = {TW, TA, TG, TV};
            const int lda[4] = {pw, pa, pg, pv};
            const uint16_t* W[4] = {y.w2, y.a2, y.g2, y.v2};
            float* Cc[4] = {L2 + 0, L2 + C, L2 + 2 * C, L2 + 3 * C};
            const int ldc[4] = {4 * C, 4 * C, 4 * C, 4 * C};
            const int N[4] = {C, C, C, C};
            const int K[4] = {pw, pa, pg, pv};
            HIPCHK(cz_launch_dec_gemm_group(st, y.has_v ? 4 : 3, A, lda, W, Cc, ldc, N, K, B));
        }
        s->prof_end(2.0 * B * (double)C * (pw + pa + pg + (y.has_v ? pv : 0)), 2.0 * C * (pw + pa + pg + pv));
        s->prof_begin(PK_RW_STATE);
        hipLaunchKernelGGL(rw_state_kernel, dim3(B, H), dim3(64), (size_t)g_rw_state_pad, st, m->proj, ldp, L2, y, l, C, S, m->v_first, m->att, m->mask, Bc, 1);
        HIPCHK(hipGetLastError());
        s->prof_end(0, st_bytes);
        s->prof_begin(PK_RW_GEMM);
        HIPCHK(cz_launch_dec_gemm_plain(st, m->att, C, y.o, m->x, C, m->x, B, C, C, nullptr, 0.0f));          // x = x + att
        s->prof_end(2.0 * B * (double)C * C, 2.0 * C * C);
        s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_prep_ffn_kernel, dim3(B), dim3(C), shm, st, m->x, y, c.norm_eps, fxp, m->xn, m->mask, 1);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * C * 4 * 4);
        s->prof_begin(PK_RW_GEMM);
        HIPCHK(cz_launch_dec_gemm_relusq(st, m->xn, C, y.fk, m->ffk, F, B, F, C));
        HIPCHK(cz_launch_dec_gemm_plain(st, m->ffk, F, y.fv, m->x, C, m->x, B, C, F, nullptr, 0.0f));          // x = x + ffn
        s->prof_end(4.0 * B * (double)C * F, 4.0 * C * F);
    }
    s->prof_begin(PK_RW_PREP);
    hipLaunchKernelGGL(rw_ln_out_kernel, dim3(B), dim3(C), shm, st, m->x, m->lno_w, m->lno_b, c.norm_eps, m->xn, 1);
    HIPCHK(hipGetLastError());
    s->prof_end(0, (double)B * C * 8);
    s->prof_begin(PK_DEC_GEMM_HEAD);
    HIPCHK(cz_launch_dec_gemm_masked(st, m->xn, C, m->head, m->logits, c.vocab, nullptr, B, c.vocab, C, m->mask));
    s->prof_end(2.0 * B * (double)c.vocab * C, 2.0 * c.vocab * C + (double)B * c.vocab * 4);
    return CZ_OK;
}

// The same step for ONE WIDE lock-step group (B >= rwkv_wide_min streams): natural-layout activations and the LDS-tiled MFMA GEMM for
// every projection instead of dec_gemm_kernel's 16-row blocks (which re-read the activations once per 32 output columns); the LoRA
// activations share one [B][320] row so that the four up-projections read slices of it.  Same canonical arithmetic, same bits.
static cz_status rw_step_wide(cz_session* s, int B, const bool gv = false) {     // gv: a handful of streams on the VALU projections (kernels_gemv.hip)
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const int C = c.hidden, F = c.d_ffn, H = C / 64, Bc = m->B;
    const int pw = pad64(c.decay_rank), pa = pad64(c.a_rank), pg = 128, pv = pad64(c.v_rank);
    const int ow = 3 * C, oa = ow + pw, og = oa + pa, ov = og + pg, ldp = ov + pv;
    const int ng = pad64(c.gate_rank);                 // rows of g1 / columns of g2 that exist in the padded weight copies (the activation slot is 128 wide)
    hipStream_t st = s->stream;
    const size_t shm = (size_t)(C + 64) * sizeof(float);
    float *TW = m->lora1a, *TA = TW + (size_t)Bc * 64, *TG = TA + (size_t)Bc * 64, *TV = TG + (size_t)Bc * 128;
    auto gemm = [&](const float* A, int lda, const uint16_t* W, float* Cc, int ldc, const float* R, int N, int K) -> hipError_t {
        s->prof_begin(PK_RW_GEMM);
        const hipError_t e = cz_launch_gemm(st, A, lda, W, Cc, ldc, R, ldc, B, N, K, B > 64 ? 1 : 0);
        s->prof_end(2.0 * B * (double)N * K, 2.0 * N * K + (double)B * (N + K) * 4);
        return e;
    };
    const double st_bytes = (double)B * H * 2.0 * 64 * 64 * 4;
    const bool mid = !gv && s->rwkv_wide_mode == 1;   // 1: the K-group GEMM (32 x 32 tiles, grouped launches); 0: the tiled big-M kernel, one launch per projection
    s->prof_begin(PK_RW_PREP);
    HIPCHK(cz_launch_embed(st, m->tok, m->emb, m->x, B, C));
    for (int l = 0; l < c.n_layers; ++l) {
        const RwLayer& y = m->L[l];
        float* axp = m->att_x_prev + (size_t)l * Bc * C; float* fxp = m->ffn_x_prev + (size_t)l * Bc * C;
        float* S = m->att_state + (size_t)l * Bc * H * 64 * 64;
        if (l) s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_prep_att_kernel, dim3(B), dim3(C), shm, st, m->x, y, l == 0 ? 1 : 0, c.norm_eps, axp, m->mix, Bc, m->mask, 0);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * C * 4 * 9);
        auto mixp = [&](int j) { return m->mix + (size_t)j * Bc * C; };                     // [6][Bc][C]: r w k v a g
        if (mid) {   // seven projections of the six mixed inputs in ONE launch of 32 x 32 tiles (r, k, v and the LoRA down-projections w1, a1, g1 (, v1))
            const float* A[7] = {mixp(0), mixp(2), mixp(3), mixp(1), mixp(4), mixp(5), mixp(3)};
            const int lda[7] = {C, C, C, C, C, C, C};
            const uint16_t* W[7] = {y.r_2, y.k_2, y.v_2, y.w1_2, y.a1_2, y.g1_2, y.v1_2};
            float* Cc[7] = {m->proj + 0, m->proj + C, m->proj + 2 * C, m->proj + ow, m->proj + oa, m->proj + og, m->proj + ov};
            const int ldc[7] = {ldp, ldp, ldp, ldp, ldp, ldp, ldp};
            const int N[7] = {C, C, C, pw, pa, ng, pv};
            const int K[7] = {C, C, C, C, C, C, C};
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_gemm_mid_group(st, y.has_v ? 7 : 6, A, lda, W, Cc, ldc, N, K, B));
            s->prof_end(2.0 * B * (double)C * (3 * C + pw + pa + pg + (y.has_v ? pv : 0)), 2.0 * C * (3 * C + pw + pa + pg + pv) + (double)B * C * 4 * 10);
        } else if (gv) {   // the same seven projections, one launch of one-wave blocks (natural-layout weights)
            const float* A[7] = {mixp(0), mixp(2), mixp(3), mixp(1), mixp(4), mixp(5), mixp(3)};
            const int lda[7] = {C, C, C, C, C, C, C};
            const uint16_t* W[7] = {y.r_n, y.k_n, y.v_n, y.w1_n, y.a1_n, y.g1_n, y.v1_n};
            float* Cc[7] = {m->proj + 0, m->proj + C, m->proj + 2 * C, m->proj + ow, m->proj + oa, m->proj + og, m->proj + ov};
            const int ldc[7] = {ldp, ldp, ldp, ldp, ldp, ldp, ldp};
            const int N[7] = {C, C, C, pw, pa, ng, pv};
            const int K[7] = {C, C, C, C, C, C, C};
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_dec_gemv_group(st, y.has_v ? 7 : 6, A, lda, W, Cc, ldc, N, K, B));
            s->prof_end(2.0 * B * (double)C * (3 * C + pw + pa + pg + (y.has_v ? pv : 0)), 2.0 * C * (3 * C + pw + pa + pg + pv) + (double)B * C * 4 * 10);
        } else {
        HIPCHK(gemm(mixp(0), C, y.r_n, m->proj + 0, ldp, nullptr, C, C));
        HIPCHK(gemm(mixp(2), C, y.k_n, m->proj + C, ldp, nullptr, C, C));
        HIPCHK(gemm(mixp(3), C, y.v_n, m->proj + 2 * C, ldp, nullptr, C, C));
        HIPCHK(gemm(mixp(1), C, y.w1_n, m->proj + ow, ldp, nullptr, pw, C));
        HIPCHK(gemm(mixp(4), C, y.a1_n, m->proj + oa, ldp, nullptr, pa, C));
        HIPCHK(gemm(mixp(5), C, y.g1_n, m->proj + og, ldp, nullptr, ng, C));
        if (y.has_v) HIPCHK(gemm(mixp(3), C, y.v1_n, m->proj + ov, ldp, nullptr, pv, C));
        }
        s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_lora_act_kernel, dim3(B), dim3(128), 0, st, m->proj, ldp, ow, oa, og, ov, c.decay_rank, c.a_rank, c.gate_rank, y.has_v ? c.v_rank : 0,
                           TW, TA, TG, TV, pw, pa, pg, pv, 0);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * (pw + pa + pg + pv) * 8);
        float* L2 = m->lora2;                                                              // [Bc][4][C]: w_lora, a_lin, g, nu_lin
        if (mid) {   // the four LoRA up-projections in one launch
            const float* A[4] = {TW, TA, TG, TV};
            const int lda[4] = {pw, pa, pg, pv};
            const uint16_t* W[4] = {y.w2_2, y.a2_2, y.g2_2, y.v2_2};
            float* Cc[4] = {L2 + 0, L2 + C, L2 + 2 * C, L2 + 3 * C};
            const int ldc[4] = {4 * C, 4 * C, 4 * C, 4 * C};
            const int N[4] = {C, C, C, C};
            const int K[4] = {pw, pa, ng, pv};
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_gemm_mid_group(st, y.has_v ? 4 : 3, A, lda, W, Cc, ldc, N, K, B));
            s->prof_end(2.0 * B * (double)C * (pw + pa + pg + (y.has_v ? pv : 0)), 2.0 * C * (pw + pa + pg + pv) + (double)B * C * 4 * 5);
        } else if (gv) {
            const float* A[4] = {TW, TA, TG, TV};
            const int lda[4] = {pw, pa, pg, pv};
            const uint16_t* W[4] = {y.w2_n, y.a2_n, y.g2_n, y.v2_n};
            float* Cc[4] = {L2 + 0, L2 + C, L2 + 2 * C, L2 + 3 * C};
            const int ldc[4] = {4 * C, 4 * C, 4 * C, 4 * C};
            const int N[4] = {C, C, C, C};
            const int K[4] = {pw, pa, ng, pv};
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_dec_gemv_group(st, y.has_v ? 4 : 3, A, lda, W, Cc, ldc, N, K, B));
            s->prof_end(2.0 * B * (double)C * (pw + pa + pg + (y.has_v ? pv : 0)), 2.0 * C * (pw + pa + pg + pv) + (double)B * C * 4 * 5);
        } else {
        HIPCHK(gemm(TW, pw, y.w2_n, L2 + 0, 4 * C, nullptr, C, pw));
        HIPCHK(gemm(TA, pa, y.a2_n, L2 + C, 4 * C, nullptr, C, pa));
        HIPCHK(gemm(TG, pg, y.g2_n, L2 + 2 * C, 4 * C, nullptr, C, ng));
        if (y.has_v) HIPCHK(gemm(TV, pv, y.v2_n, L2 + 3 * C, 4 * C, nullptr, C, pv));
        }
        s->prof_begin(PK_RW_STATE);
        hipLaunchKernelGGL(rw_state_kernel, dim3(B, H), dim3(64), (size_t)g_rw_state_pad, st, m->proj, ldp, L2, y, l, C, S, m->v_first, m->att, m->mask, Bc, 0);
        HIPCHK(hipGetLastError());
        s->prof_end(0, st_bytes);
        if (mid) {
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_gemm_mid(st, m->att, C, y.o_2, m->x, C, m->x, C, B, C, C));
            s->prof_end(2.0 * B * (double)C * C, 2.0 * C * C + (double)B * C * 12);
        } else if (gv) {
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_dec_gemv_plain(st, m->att, C, y.o_n, m->x, C, m->x, B, C, C, nullptr, 0.0f));
            s->prof_end(2.0 * B * (double)C * C, 2.0 * C * C + (double)B * C * 12);
        } else HIPCHK(gemm(m->att, C, y.o_n, m->x, C, m->x, C, C));                        // x = x + att
        s->prof_begin(PK_RW_PREP);
        hipLaunchKernelGGL(rw_prep_ffn_kernel, dim3(B), dim3(C), shm, st, m->x, y, c.norm_eps, fxp, m->xn, m->mask, 0);
        HIPCHK(hipGetLastError());
        s->prof_end(0, (double)B * C * 4 * 4);
        s->prof_begin(PK_RW_GEMM);
        if (mid) HIPCHK(cz_launch_gemm_mid_relusq(st, m->xn, C, y.fk_2, m->ffk, F, B, F, C));
        else if (gv) HIPCHK(cz_launch_dec_gemv_relusq(st, m->xn, C, y.fk_n, m->ffk, F, B, F, C));
        else HIPCHK(cz_launch_gemm_relusq(st, m->xn, C, y.fk_n, m->ffk, F, B, F, C));
        s->prof_end(2.0 * B * (double)C * F, 2.0 * C * F + (double)B * (C + F) * 4);
        if (mid) {
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_gemm_mid(st, m->ffk, F, y.fv_2, m->x, C, m->x, C, B, C, F));
            s->prof_end(2.0 * B * (double)C * F, 2.0 * C * F + (double)B * (C + F) * 4);
        } else if (gv) {
            s->prof_begin(PK_RW_GEMM);
            HIPCHK(cz_launch_dec_gemv_plain(st, m->ffk, F, y.fv_n, m->x, C, m->x, B, C, F, nullptr, 0.0f));
            s->prof_end(2.0 * B * (double)C * F, 2.0 * C * F + (double)B * (C + F) * 4);
        } else HIPCHK(gemm(m->ffk, F, y.fv_n, m->x, C, m->x, C, F));                       // x = x + ffn
    }
    s->prof_begin(PK_RW_PREP);
    hipLaunchKernelGGL(rw_ln_out_kernel, dim3(B), dim3(C), shm, st, m->x, m->lno_w, m->lno_b, c.norm_eps, m->xn, 0);
    HIPCHK(hipGetLastError());
    s->prof_end(0, (double)B * C * 8);
    s->prof_begin(PK_DEC_GEMM_HEAD);
    if (gv) HIPCHK(cz_launch_dec_gemv_masked(st, m->xn, C, m->head_n, m->logits, c.vocab, B, c.vocab, C, m->mask));
    else HIPCHK(cz_launch_gemm_masked(st, m->xn, C, m->head_n, m->logits, c.vocab, B, c.vocab, C, m->mask));
    s->prof_end(2.0 * B * (double)c.vocab * C, 2.0 * c.vocab * C + (double)B * c.vocab * 4);
    return CZ_OK;
}
static cz_status rw_step(cz_session* s, int B) {
    if (s->rwkv_wide_min > 0 && B >= s->rwkv_wide_min) return rw_step_wide(s, B);
    {   // a handful of streams (one file = one stream): the wide step's natural-layout dataflow on the one-wave VALU projections
        const cz_rwkv7_config& c = s->rwkv->cfg;
        // (up to four streams here: at the 0.1B shape — K = 768 and 3072 — eight tie with dec_gemm_kernel, sixteen lose 18 %: profiles/r04_single_stream/)
        if (B <= 4 && cz_dec_gemv_ok(B, c.hidden, false) && cz_dec_gemv_ok(B, c.d_ffn, false) && c.hidden <= 768 && c.d_ffn % 64 == 0) return rw_step_wide(s, B, true);
    }
    return rw_step_narrow(s, B);
}

// ------------------------------------------------------------------ session trait (src/models.rs:147-180)
cz_status cz_rwkv7_step_host(cz_session* s, uint32_t token, float* out_logits) {
    Rwkv7Model* m = s->rwkv;
    if (token >= (uint32_t)m->cfg.vocab) { cz_set_error("token id %u >= vocab %d", token, m->cfg.vocab); return CZ_ERR_ARG; }
    const bool fresh = m->B == 0;
    CZCHK(rw_ensure_streams(s, 1));
    if (fresh) CZCHK(rw_reset(s));
    hipLaunchKernelGGL(rw_fill_kernel, dim3(1), dim3(64), 0, s->stream, m->tok, m->mask, token, 1);
    CZCHK(rw_step(s, 1));
    HIPCHK(hipMemcpyAsync(out_logits, m->logits, (size_t)m->cfg.vocab * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->index_pos += 1;
    return CZ_OK;
}
cz_status cz_rwkv7_reprime_host(cz_session* s, const uint32_t* hist, size_t n, float* out_logits) {   // :161-180: reset, then feed one by one
    Rwkv7Model* m = s->rwkv;
    for (size_t i = 0; i < n; ++i) if (hist[i] >= (uint32_t)m->cfg.vocab) { cz_set_error("token id %u >= vocab", hist[i]); return CZ_ERR_ARG; }
    CZCHK(rw_ensure_streams(s, 1));
    CZCHK(rw_reset(s));
    s->index_pos = 0;
    for (size_t i = 0; i < n; ++i) {
        hipLaunchKernelGGL(rw_fill_kernel, dim3(1), dim3(64), 0, s->stream, m->tok, m->mask, hist[i], 1);
        CZCHK(rw_step(s, 1));
        s->index_pos += 1;
    }
    HIPCHK(hipMemcpyAsync(out_logits, m->logits, (size_t)m->cfg.vocab * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CZ_OK;
}

// ------------------------------------------------------------------ token loops, RWKV branch
// capture one iteration of a lock-step loop (`body` enqueues it on s->stream).  Every buffer the RWKV step touches is allocated
// once (rw_ensure_streams), so only the session's DevBufs can move: s->devbuf_gen is part of the key.
template <class F>
static cz_status rw_graph_build(cz_session* s, Rwkv7Model::Graph& G, int B, F body) {
    rw_graph_destroy(G);
    const bool prof = s->prof_on; s->prof_on = false;            // no events inside a captured region
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    const cz_status rc = body();
    hipGraph_t g = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s->stream, &g);
    s->prof_on = prof;
    if (rc != CZ_OK || e2 != hipSuccess || !g) {
        if (g) hipGraphDestroy(g);
        if (rc == CZ_OK) cz_set_error("hipGraph capture of the RWKV step failed (%s)", hipGetErrorString(e2));
        return rc != CZ_OK ? rc : CZ_ERR_HIP;
    }
    G.g = g;
    HIPCHK(hipGraphInstantiate(&G.e, G.g, nullptr, nullptr, 0));
    G.B = B; G.gen = s->devbuf_gen;
    return CZ_OK;
}

// a lane shares the parent's weights and owns its recurrent state and work buffers (allocated on first use)
cz_status cz_rwkv7_make_lane(const cz_session* parent, cz_session* child) {
    const Rwkv7Model* pm = parent->rwkv;
    Rwkv7Model* m = new Rwkv7Model();
    m->cfg = pm->cfg; m->emb = pm->emb; m->head = pm->head; m->head_n = pm->head_n; m->lno_w = pm->lno_w; m->lno_b = pm->lno_b; m->L = pm->L;
    child->rwkv = m;                                               // m->allocs stays empty: cz_rwkv7_destroy frees the lane's own buffers only
    return CZ_OK;
}

// decode_symbol(i) for every stream, then (optionally) the model step on the decoded symbols
static cz_status rw_decode_iteration(cz_session* s, int B, unsigned long long i, const unsigned long long* step_dev, bool step_after) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    cz_dec_state* dst = (cz_dec_state*)s->decst.p;
    s->prof_begin(PK_DECODE);
    HIPCHK(cz_launch_decode_symbol(s->stream, m->logits, c.vocab, dst, m->sym, B, c.vocab, 1, i, s->fallback_dev, step_dev, s->cdf_scratch));
    s->prof_end(0, (double)B * c.vocab * 4);
    if (step_after) {
        hipLaunchKernelGGL(rw_mask_from_sym_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, dst, m->sym, (uint32_t)c.vocab, m->tok, m->mask, B, i, step_dev);
        CZCHK(rw_step(s, B));
    }
    return CZ_OK;
}

// one lock-step group of B <= max_streams streams
cz_status cz_rwkv7_encode_group(cz_session* s, int B, const uint32_t* const* ids, const size_t* n, uint32_t* const* lo, uint32_t* const* hi) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const uint32_t V = (uint32_t)c.vocab;
    {
        const size_t g0 = 0;
        CZCHK(rw_ensure_streams(s, B));
        CZCHK(rw_reset(s));
        std::vector<unsigned long long> off(B), len(B); size_t total = 0, maxn = 0;
        for (int b = 0; b < B; ++b) {
            if (n[g0 + b] == 0) { cz_set_error("stream %zu is empty (ids must start with BOS)", g0 + b); return CZ_ERR_ARG; }
            for (size_t i = 0; i < n[g0 + b]; ++i) if (ids[g0 + b][i] >= V + 256) { cz_set_error("symbol %u >= vocab + 256", ids[g0 + b][i]); return CZ_ERR_ARG; }
            if (ids[g0 + b][0] >= V) { cz_set_error("BOS must be a vocabulary token"); return CZ_ERR_ARG; }
            off[b] = total; len[b] = n[g0 + b]; total += n[g0 + b]; maxn = std::max(maxn, n[g0 + b]);
        }
        std::vector<uint32_t> flat(total);
        for (int b = 0; b < B; ++b) memcpy(flat.data() + off[b], ids[g0 + b], n[g0 + b] * 4);
        CZCHK(s->tok.ensure(total * 4)); CZCHK(s->seqd.ensure(B * 16)); CZCHK(s->roped.ensure(B * 16));
        const size_t npairs = (size_t)B * (maxn ? maxn - 1 : 0);
        CZCHK(s->clo.ensure(npairs * 4 + 16)); CZCHK(s->chi.ensure(npairs * 4 + 16));
        HIPCHK(hipMemcpyAsync(s->tok.p, flat.data(), total * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->seqd.p, off.data(), B * 8, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->roped.p, len.data(), B * 8, hipMemcpyHostToDevice, s->stream));
        const bool use_graph = !s->no_graph && !s->prof_on && cz_cdf_par_supported(c.vocab, 1);
        auto iteration = [&](unsigned long long i, const unsigned long long* step_dev) -> cz_status {
            hipLaunchKernelGGL(rw_pick_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, (const uint32_t*)s->tok.p, (const unsigned long long*)s->seqd.p,
                               (const unsigned long long*)s->roped.p, i, V, m->tok, m->sym, m->mask, B, step_dev);
            CZCHK(rw_step(s, B));
            s->prof_begin(PK_CDF);
            HIPCHK(cz_launch_cdf_pairs(s->stream, m->logits, c.vocab, m->sym, (uint32_t*)s->clo.p + (step_dev ? 0 : i * B), (uint32_t*)s->chi.p + (step_dev ? 0 : i * B), B,
                                       c.vocab, 1, s->fallback_dev, s->cdf_scratch, step_dev));
            s->prof_end(0, (double)B * c.vocab * 4);
            return CZ_OK;
        };
        if (use_graph) {
            if (!m->genc.e || m->genc.B != B || m->genc.gen != s->devbuf_gen)
                CZCHK(rw_graph_build(s, m->genc, B, [&]() -> cz_status {
                    CZCHK(iteration(0, m->step_dev));
                    hipLaunchKernelGGL(rw_counter_kernel, dim3(1), dim3(1), 0, s->stream, m->step_dev, 1);
                    return CZ_OK;
                }));
            hipLaunchKernelGGL(rw_counter_kernel, dim3(1), dim3(1), 0, s->stream, m->step_dev, 0);
        }
        for (size_t i = 0; i + 1 < maxn; ++i) {
            if (use_graph) HIPCHK(hipGraphLaunch(m->genc.e, s->stream));
            else CZCHK(iteration((unsigned long long)i, nullptr));
            if (s->step_sync || (i % CZ_SYNC_EVERY) == CZ_SYNC_EVERY - 1) HIPCHK(hipStreamSynchronize(s->stream));
        }
        std::vector<uint32_t> hlo(npairs), hhi(npairs);
        if (npairs) {
            HIPCHK(hipMemcpyAsync(hlo.data(), s->clo.p, npairs * 4, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(hhi.data(), s->chi.p, npairs * 4, hipMemcpyDeviceToHost, s->stream));
        }
        HIPCHK(hipStreamSynchronize(s->stream));
        for (int b = 0; b < B; ++b)
            for (size_t i = 0; i + 1 < n[g0 + b]; ++i) { lo[g0 + b][i] = hlo[i * B + b]; hi[g0 + b][i] = hhi[i * B + b]; }
    }
    return CZ_OK;
}

// ------------------------------------------------------------------ sequence-parallel encode
// Encode is teacher-forced: every token a stream will feed the model is known up front.  RWKV7's only recurrence in t is the WKV state
// S_t = f(S_{t-1}, w_t, k_t, v_t, a_t) (rwkv7.rs:289-301) and the one-token shift of the LayerNorm'd input (:193-203, :414-425) — and
// the latter needs x_{t-1} of the SAME layer, which is available for all t as soon as the layer below has run for all t.  So per
// layer ALL T tokens of all streams go through the token-shift mixes and every projection as big-M GEMMs on the tiled MFMA kernel
// (bit-identical to the T == 1 step: an output element's reduction order depends on K only), the state-independent per-token terms
// (decay w, in-context rate a, normalised kk, k, the v-residual mix) are evaluated in parallel over (row, head), and only the scan
// S_t, y_t = S_t r_t runs sequentially in t: one wave per (stream, head) with its 64 x 64 state tile in registers for the whole block —
// the state never touches HBM between tokens (the lock-step step moves 2 x 16 KB per head, layer and token).  GroupNorm, bonus and gate
// of y_t do not feed the recurrence and run in parallel again.  Literal symbols (id >= vocab) are coded but never step the model
// (src/main.rs:2347): the model's row sequence is the stream's tokens with the literals removed, and a coded symbol reads the logits of
// the last row at or before it.
struct RwMixW { const uint16_t* w[6]; };
// rows [M][C]: Y[r] = LayerNorm(X[src(r)]) ; pre_w != nullptr (layer 0): X[r] = pre-LayerNorm(X[r]) in place first (rwkv7.rs:468-470)
__global__ void rw_seq_ln_kernel(float* __restrict__ X, const int* __restrict__ rows, const uint16_t* pre_w, const uint16_t* pre_b, const uint16_t* w, const uint16_t* b,
                                 float eps, float* __restrict__ Y) {
    extern __shared__ float sh[];
    const int r = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const size_t src = rows ? (size_t)rows[r] : (size_t)r;
    float x = X[src * C + tid];
    if (pre_w) { x = block_layer_norm(x, pre_w, pre_b, eps, sh); X[src * C + tid] = x; }
    Y[(size_t)r * C + tid] = block_layer_norm(x, w, b, eps, sh);
}
// token-shift mixes (rwkv7.rs:193-203 / :419-424): out_j[r] = xa + (prev - xa) * w_j, prev = the LayerNorm'd row before r in its stream:
// row_prev[r] >= 0 -> XA[row_prev[r]] (inside this block), < 0 -> the stream's carry row -1 - row_prev[r] (state.att_x_prev / ffn_x_prev)
template <int NMIX>
__global__ void rw_seq_mix_kernel(const float* __restrict__ XA, const int* __restrict__ row_prev, const float* __restrict__ carry, RwMixW mw,
                                  float* __restrict__ MIX, size_t mix_stride) {
    const int r = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const float xa = XA[(size_t)r * C + tid];
    const int pr = row_prev[r];
    const float prev = pr >= 0 ? XA[(size_t)pr * C + tid] : carry[(size_t)(-1 - pr) * C + tid];
    const float xx = prev - xa;
#pragma unroll
    for (int j = 0; j < NMIX; ++j) MIX[(size_t)j * mix_stride + (size_t)r * C + tid] = xa + xx * cz_bf16_to_f32(mw.w[j][tid]);
}
// carry[stream] = XA[last row of the stream in this block]
__global__ void rw_seq_carry_kernel(const float* __restrict__ XA, const int* __restrict__ s_row0, const int* __restrict__ s_len, const int* __restrict__ s_stream,
                                    float* __restrict__ carry) {
    const int k = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    carry[(size_t)s_stream[k] * C + tid] = XA[(size_t)(s_row0[k] + s_len[k] - 1) * C + tid];
}
// LoRA hidden activations (rwkv7.rs:211, 229, 235, 244), natural layout, zero-padded to the 64-multiples the GEMMs take: TL[r] = tanh(w1.) | a1. | sigmoid(g1.) | v1.
__global__ void rw_seq_lora_act_kernel(const float* __restrict__ P, int ldp, int ow, int oa, int og, int ov, int rw, int ra, int rg, int rv,
                                       float* __restrict__ TL, int pw, int pa, int pg, int pv) {
    const int r = blockIdx.x, t = threadIdx.x;
    const float* p = P + (size_t)r * ldp;
    float* tl = TL + (size_t)r * (pw + pa + pg + pv);
    if (t < pw) tl[t] = t < rw ? cz_tanhf(p[ow + t]) : 0.0f;
    if (t < pa) tl[pw + t] = t < ra ? p[oa + t] : 0.0f;
    if (t < pg) tl[pw + pa + t] = t < rg ? cz_sigmoidf(p[og + t]) : 0.0f;
    if (t < pv) tl[pw + pa + pg + t] = t < rv ? p[ov + t] : 0.0f;
}
// per (row, head): everything of the time-mix that does not depend on the state (rwkv7.rs:213-266): SC[r][h] = kk | kk*a | w | k | r | v  (6 x 64 floats)
#define RW_SC 384
__global__ __launch_bounds__(64) void rw_seq_pre_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ L2 /*[M][4C]: w a g nu*/, const RwLayer L, int layer, int C,
                                                        float* __restrict__ VF /*[M][C]: layer 0's v*/, float* __restrict__ SC) {
    __shared__ float kk[64];
    const int r = blockIdx.x, h = blockIdx.y, i = threadIdx.x, c = h * 64 + i, H = gridDim.y;
    const float* p = P + (size_t)r * ldp;
    const float* l2 = L2 + (size_t)r * 4 * C;
    const float rr = p[c], kraw = p[C + c];
    float v = p[2 * C + c];
    const float wc = l2[c] + cz_bf16_to_f32(L.w0[c]);
    const float sp = cz_logf(cz_expf(-wc) + 1.0f);
    const float wt = (-sp) + (-0.5f);
    const float w = cz_expf(-cz_expf(wt));
    const float a = cz_sigmoidf(l2[C + c] + cz_bf16_to_f32(L.a0[c]));
    if (layer == 0) VF[(size_t)r * C + c] = v;
    else if (L.has_v) {
        const float nu = cz_sigmoidf(l2[3 * C + c] + cz_bf16_to_f32(L.v0[c]));
        const float dl = VF[(size_t)r * C + c] - v;
        v = v + dl * nu;
    }
    const float kk0 = kraw * cz_bf16_to_f32(L.k_k[c]);
    kk[i] = kk0;
    __syncthreads();
    float ss = 0.0f;
    for (int j = 0; j < 64; ++j) { const float q = kk[j] * kk[j]; ss = ss + q; }
    float nrm = sqrtf(ss);
    if (!(nrm > 1e-12f)) nrm = 1e-12f;
    const float kkn = kk0 / nrm;
    const float sc = 1.0f + (a - 1.0f) * cz_bf16_to_f32(L.k_a[c]);
    const float k = kraw * sc;
    float* o = SC + ((size_t)r * H + h) * RW_SC;
    o[i] = kkn; o[64 + i] = kkn * a; o[128 + i] = w; o[192 + i] = k; o[256 + i] = rr; o[320 + i] = v;
}
// the scan: block = (active stream, head), one wave, lane i = state row i held in 64 registers for the stream's whole block of rows
// (rwkv7.rs:289-301): t = S kk ; S' = S diag(w) - t (kk a)^T + v k^T ; y = S' r.  The next row's six vectors are in flight while a row is computed.
__global__ __launch_bounds__(64) void rw_seq_scan_kernel(const float* __restrict__ SC, const int* __restrict__ s_row0, const int* __restrict__ s_len, const int* __restrict__ s_stream,
                                                         float* __restrict__ S, float* __restrict__ Y, int C) {
    __shared__ __attribute__((aligned(16))) float bc[2][5][64];
    const int k = blockIdx.x, h = blockIdx.y, i = threadIdx.x, H = gridDim.y;
    const int b = s_stream[k], row0 = s_row0[k], len = s_len[k];
    float* Srow = S + ((size_t)b * H + h) * 4096 + (size_t)i * 4;          // tiled state (see rw_state_kernel)
    float srow[64];
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
        const float4 t4 = *reinterpret_cast<const float4*>(Srow + 256 * j4);
        srow[4 * j4] = t4.x; srow[4 * j4 + 1] = t4.y; srow[4 * j4 + 2] = t4.z; srow[4 * j4 + 3] = t4.w;
    }
    const float* sc = SC + ((size_t)row0 * H + h) * RW_SC + i;
    float nx[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) nx[q] = sc[64 * q];
    for (int t = 0; t < len; ++t) {
        float cur[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) cur[q] = nx[q];
        if (t + 1 < len) {
            const float* sn_ = sc + (size_t)(t + 1) * H * RW_SC;
#pragma unroll
            for (int q = 0; q < 6; ++q) nx[q] = sn_[64 * q];
        }
        float (*B)[64] = bc[t & 1];
#pragma unroll
        for (int q = 0; q < 5; ++q) B[q][i] = cur[q];
        __syncthreads();                                                   // (one wave: the LDS writes above are visible to every lane's reads below)
        const float v = cur[5];
        float tt = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; ++j) tt = cz_fma(srow[j], B[0][j], tt);                           // (S @ kk)[i]
        float y = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const float sn = (srow[j] * B[2][j] - tt * B[1][j]) + v * B[3][j];
            srow[j] = sn;
            y = cz_fma(sn, B[4][j], y);                                                           // (S' @ r)[i]
        }
        Y[(size_t)(row0 + t) * C + h * 64 + i] = y;
    }
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) *reinterpret_cast<float4*>(Srow + 256 * j4) = make_float4(srow[4 * j4], srow[4 * j4 + 1], srow[4 * j4 + 2], srow[4 * j4 + 3]);
}
// per (row, head): GroupNorm of y (rwkv7.rs:303, 529-542), bonus (r . k r_k) v (:305-321), gate (:323) -> OUT[r][c], natural layout
__global__ __launch_bounds__(64) void rw_seq_post_kernel(const float* __restrict__ Y, const float* __restrict__ SC, const float* __restrict__ L2, const RwLayer L, int C,
                                                         float* __restrict__ OUT) {
    __shared__ float yv[64], rv[64], kv[64];
    const int r = blockIdx.x, h = blockIdx.y, i = threadIdx.x, c = h * 64 + i, H = gridDim.y;
    const float* sc = SC + ((size_t)r * H + h) * RW_SC;
    const float y = Y[(size_t)r * C + c], v = sc[320 + i], g = L2[(size_t)r * 4 * C + 2 * C + c];
    yv[i] = y; rv[i] = sc[256 + i]; kv[i] = sc[192 + i];
    __syncthreads();
    float s1 = 0.0f;
    for (int j = 0; j < 64; ++j) s1 = s1 + yv[j];
    const float mean = s1 * 0.015625f;
    const float cen = y - mean;
    __syncthreads();
    yv[i] = cen * cen;
    __syncthreads();
    float s2 = 0.0f, al = 0.0f;
    for (int j = 0; j < 64; ++j) s2 = s2 + yv[j];
    for (int j = 0; j < 64; ++j) { const float q = rv[j] * kv[j] * cz_bf16_to_f32(L.r_k[h * 64 + j]); al = al + q; }
    const float den = sqrtf(s2 * 0.015625f + (float)64e-5);
    const float gn = cen / den * cz_bf16_to_f32(L.gn_w[c]) + cz_bf16_to_f32(L.gn_b[c]);
    const float ad = al * v;
    OUT[(size_t)r * C + c] = (gn + ad) * g;
}

static const double P_MIN_HOST = 1.862645149230957e-09;      // ac_p_min() = 2 * 2^-30 (src/main.rs:235-238; P_MIN of cz_cdf_common.h)
int g_rw_seq_rows = 32768;     // tuning knob (cz_debug_set "rwkv_seq_rows"): rows (model steps of all streams) per time block of the sequence-parallel encode

// Streams [0, B) of one group (B <= max_streams: the recurrent state arrays have that many slots).  Coder intervals straight into lo / hi.
// xe form (first_sym / p_out given): only the coded symbols i >= first_sym[b] are evaluated, and what comes back is the probability
// combined_pdf_with_literals gives each of them (p_out[b][i - first_sym[b]]) instead of its coder interval.
static cz_status rw_encode_seq_group(cz_session* s, int B, const uint32_t* const* ids, const size_t* n, uint32_t* const* lo, uint32_t* const* hi,
                                     const size_t* first_sym = nullptr, double* const* p_out = nullptr) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const uint32_t V = (uint32_t)c.vocab;
    const int C = c.hidden, F = c.d_ffn, H = C / 64, Bc = s->max_streams, L = c.n_layers;
    const int pw = pad64(c.decay_rank), pa = pad64(c.a_rank), pg = pad64(c.gate_rank), pv = pad64(c.v_rank), ptl = pw + pa + pg + pv;
    const int ow = 3 * C, oa = ow + pw, og = oa + pa, ov = og + pg, ldp = ov + pv;
    hipStream_t st = s->stream;
    CZCHK(rw_ensure_streams(s, B));
    CZCHK(rw_reset(s));
    // the model's row sequence per stream, and for every coded symbol i the row whose logits code it
    std::vector<std::vector<uint32_t>> toks(B); std::vector<std::vector<uint32_t>> rowof(B);
    size_t max_steps = 0;
    for (int b = 0; b < B; ++b) {
        if (n[b] == 0) { cz_set_error("stream %d is empty (ids must start with BOS)", b); return CZ_ERR_ARG; }
        for (size_t i = 0; i < n[b]; ++i) if (ids[b][i] >= V + 256) { cz_set_error("symbol %u >= vocab + 256", ids[b][i]); return CZ_ERR_ARG; }
        if (ids[b][0] >= V) { cz_set_error("BOS must be a vocabulary token"); return CZ_ERR_ARG; }
        rowof[b].resize(n[b] - 1);
        for (size_t i = 0; i + 1 < n[b]; ++i) {
            if (i == 0 || ids[b][i] < V) toks[b].push_back(ids[b][i]);                       // (:2347: literals are coded, the model is not stepped)
            rowof[b][i] = (uint32_t)(toks[b].size() - 1);
        }
        max_steps = std::max(max_steps, toks[b].size());
    }
    if (max_steps == 0) return CZ_OK;
    const size_t Tb = std::max<size_t>(1, (size_t)g_rw_seq_rows / (size_t)B);
    const size_t shm = (size_t)(C + 64) * sizeof(float);
    std::vector<size_t> sym_cur(B, 0);                                                       // next coded symbol of every stream
    for (size_t t0 = 0; t0 < max_steps; t0 += Tb) {
        // pack the block: rows of stream b = its steps [t0, min(t0 + Tb, steps_b))
        std::vector<int> s_row0, s_len, s_stream, row_prev; std::vector<uint32_t> tok;
        for (int b = 0; b < B; ++b) {
            if (toks[b].size() <= t0) continue;
            const size_t len = std::min(Tb, toks[b].size() - t0);
            s_row0.push_back((int)tok.size()); s_len.push_back((int)len); s_stream.push_back(b);
            for (size_t t = 0; t < len; ++t) { row_prev.push_back(t == 0 ? -1 - b : (int)tok.size() - 1); tok.push_back(toks[b][t0 + t]); }
        }
        const int M = (int)tok.size(), nb = (int)s_stream.size();
        // coded symbols whose logits row lies in this block
        std::vector<int> crow; std::vector<uint32_t> csym; struct Dst { int b; size_t i; }; std::vector<Dst> cdst;
        for (int k = 0; k < nb; ++k) {
            const int b = s_stream[k];
            size_t& i = sym_cur[b];
            while (i + 1 < n[b] && rowof[b][i] < t0 + (size_t)s_len[k]) {
                if (!first_sym || i >= first_sym[b]) { crow.push_back(s_row0[k] + (int)(rowof[b][i] - t0)); csym.push_back(ids[b][i + 1]); cdst.push_back({b, i}); }
                ++i;
            }
        }
        const size_t Lr = crow.size();
        // device copies of the block's index arrays: s_row0 | s_len | s_stream | row_prev | crow, then tok, csym
        const size_t n_idx = (size_t)3 * nb + M + Lr;
        CZCHK(m->q_idx.ensure((n_idx + (size_t)M + Lr) * 4 + 64));
        std::vector<int> hidx; hidx.reserve(n_idx + M + Lr);
        hidx.insert(hidx.end(), s_row0.begin(), s_row0.end()); hidx.insert(hidx.end(), s_len.begin(), s_len.end()); hidx.insert(hidx.end(), s_stream.begin(), s_stream.end());
        hidx.insert(hidx.end(), row_prev.begin(), row_prev.end()); hidx.insert(hidx.end(), crow.begin(), crow.end());
        for (uint32_t t : tok) hidx.push_back((int)t);
        for (uint32_t t : csym) hidx.push_back((int)t);
        HIPCHK(hipMemcpyAsync(m->q_idx.p, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice, st));
        const int* d_row0 = (const int*)m->q_idx.p; const int* d_len = d_row0 + nb; const int* d_stream = d_len + nb; const int* d_prev = d_stream + nb;
        const int* d_crow = d_prev + M; const uint32_t* d_tok = (const uint32_t*)(d_crow + Lr); const uint32_t* d_csym = d_tok + M;
        CZCHK(m->q_x.ensure((size_t)M * C * 4)); CZCHK(m->q_xa.ensure((size_t)M * C * 4)); CZCHK(m->q_mix.ensure((size_t)6 * M * C * 4));
        CZCHK(m->q_p.ensure((size_t)M * ldp * 4)); CZCHK(m->q_tl.ensure((size_t)M * ptl * 4)); CZCHK(m->q_l2.ensure((size_t)M * 4 * C * 4));
        CZCHK(m->q_sc.ensure((size_t)M * H * RW_SC * 4)); CZCHK(m->q_vf.ensure((size_t)M * C * 4)); CZCHK(m->q_y.ensure((size_t)M * C * 4));
        CZCHK(m->q_ffk.ensure((size_t)M * F * 4));
        float *X = (float*)m->q_x.p, *XA = (float*)m->q_xa.p, *MIX = (float*)m->q_mix.p, *P = (float*)m->q_p.p, *TL = (float*)m->q_tl.p, *L2 = (float*)m->q_l2.p;
        float *SC = (float*)m->q_sc.p, *VF = (float*)m->q_vf.p, *Y = (float*)m->q_y.p, *FFK = (float*)m->q_ffk.p;
        const size_t ms = (size_t)M * C;
        auto gemm = [&](const float* A, int lda, const uint16_t* W, float* Cc, int ldc, const float* R, int N, int K) -> hipError_t {
            s->prof_begin(M <= 64 ? PK_GEMM_SKINNY : (N % 128 ? PK_GEMM_N64 : PK_GEMM));
            const hipError_t e = cz_launch_gemm(st, A, lda, W, Cc, ldc, R, ldc, M, N, K, 0);
            s->prof_end(2.0 * M * (double)N * K, (double)N * K * 2 + (double)M * (N + K) * 4);
            return e;
        };
        s->prof_begin(PK_EMBED); HIPCHK(cz_launch_embed(st, d_tok, m->emb, X, M, C)); s->prof_end(0, (double)M * C * 6);
        for (int l = 0; l < L; ++l) {
            const RwLayer& y = m->L[l];
            float* axp = m->att_x_prev + (size_t)l * Bc * C; float* fxp = m->ffn_x_prev + (size_t)l * Bc * C;
            float* S = m->att_state + (size_t)l * Bc * H * 64 * 64;
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_ln_kernel, dim3(M), dim3(C), shm, st, X, (const int*)nullptr, l == 0 ? y.pre_w : nullptr, y.pre_b, y.ln1_w, y.ln1_b, c.norm_eps, XA);
            RwMixW mw{{y.x_r, y.x_w, y.x_k, y.x_v, y.x_a, y.x_g}};
            hipLaunchKernelGGL((rw_seq_mix_kernel<6>), dim3(M), dim3(C), 0, st, XA, d_prev, axp, mw, MIX, ms);
            hipLaunchKernelGGL(rw_seq_carry_kernel, dim3(nb), dim3(C), 0, st, XA, d_row0, d_len, d_stream, axp);      // state.att_x_prev (:308)
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * C * 4 * 10);
            // projections of the mixed inputs (order r w k v a g): r, k, v, and the LoRA down-projections
            HIPCHK(gemm(MIX + 0 * ms, C, y.r_n, P + 0, ldp, nullptr, C, C));
            HIPCHK(gemm(MIX + 2 * ms, C, y.k_n, P + C, ldp, nullptr, C, C));
            HIPCHK(gemm(MIX + 3 * ms, C, y.v_n, P + 2 * C, ldp, nullptr, C, C));
            HIPCHK(gemm(MIX + 1 * ms, C, y.w1_n, P + ow, ldp, nullptr, pw, C));
            HIPCHK(gemm(MIX + 4 * ms, C, y.a1_n, P + oa, ldp, nullptr, pa, C));
            HIPCHK(gemm(MIX + 5 * ms, C, y.g1_n, P + og, ldp, nullptr, pg, C));
            if (y.has_v) HIPCHK(gemm(MIX + 3 * ms, C, y.v1_n, P + ov, ldp, nullptr, pv, C));
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_lora_act_kernel, dim3(M), dim3(128), 0, st, P, ldp, ow, oa, og, ov, c.decay_rank, c.a_rank, c.gate_rank, y.has_v ? c.v_rank : 0, TL, pw, pa, pg, pv);
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * ptl * 8);
            HIPCHK(gemm(TL, ptl, y.w2_n, L2 + 0, 4 * C, nullptr, C, pw));
            HIPCHK(gemm(TL + pw, ptl, y.a2_n, L2 + C, 4 * C, nullptr, C, pa));
            HIPCHK(gemm(TL + pw + pa, ptl, y.g2_n, L2 + 2 * C, 4 * C, nullptr, C, pg));
            if (y.has_v) HIPCHK(gemm(TL + pw + pa + pg, ptl, y.v2_n, L2 + 3 * C, 4 * C, nullptr, C, pv));
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_pre_kernel, dim3(M, H), dim3(64), 0, st, P, ldp, L2, y, l, C, VF, SC);
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * H * RW_SC * 4 + (double)M * C * 4 * 8);
            s->prof_begin(PK_RW_SCAN);
            hipLaunchKernelGGL(rw_seq_scan_kernel, dim3(nb, H), dim3(64), 0, st, SC, d_row0, d_len, d_stream, S, Y, C);
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * H * RW_SC * 4 + (double)M * C * 4 + (double)nb * H * 2.0 * 64 * 64 * 4);
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_post_kernel, dim3(M, H), dim3(64), 0, st, Y, SC, L2, y, C, XA);                    // (XA is free again: the gated output goes there)
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * H * 3 * 64 * 4 + (double)M * C * 4 * 3);
            HIPCHK(gemm(XA, C, y.o_n, X, C, X, C, C));                                                                   // x = x + att
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_ln_kernel, dim3(M), dim3(C), shm, st, X, (const int*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)nullptr, y.ln2_w, y.ln2_b, c.norm_eps, XA);
            RwMixW fw{{y.fx_k, nullptr, nullptr, nullptr, nullptr, nullptr}};
            hipLaunchKernelGGL((rw_seq_mix_kernel<1>), dim3(M), dim3(C), 0, st, XA, d_prev, fxp, fw, MIX, ms);
            hipLaunchKernelGGL(rw_seq_carry_kernel, dim3(nb), dim3(C), 0, st, XA, d_row0, d_len, d_stream, fxp);
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)M * C * 4 * 5);
            s->prof_begin(PK_GEMM);
            HIPCHK(cz_launch_gemm_relusq(st, MIX, C, y.fk_n, FFK, F, M, F, C));
            s->prof_end(2.0 * M * (double)F * C, (double)F * C * 2 + (double)M * (F + C) * 4);
            HIPCHK(gemm(FFK, F, y.fv_n, X, C, X, C, F));                                                                 // x = x + ffn
        }
        // logits of the coded symbols' rows -> coder intervals (combined_pdf_with_literals + inline quantiser, src/main.rs:2303-2325)
        CZCHK(s->clo.ensure(Lr * (p_out ? 8 : 4) + 16)); CZCHK(s->chi.ensure(Lr * 4 + 16));
        const size_t LB = std::max<size_t>(1, std::min(s->logit_batch_rows, (size_t)(1ull << 31) / ((size_t)c.vocab * 4)));
        CZCHK(s->logits.ensure(std::min(LB, std::max<size_t>(Lr, 1)) * (size_t)c.vocab * 4)); CZCHK(m->q_xn.ensure(std::min(LB, std::max<size_t>(Lr, 1)) * (size_t)C * 4));
        for (size_t r0 = 0; r0 < Lr; r0 += LB) {
            const int nr = (int)std::min(LB, Lr - r0);
            s->prof_begin(PK_RW_PREP);
            hipLaunchKernelGGL(rw_seq_ln_kernel, dim3(nr), dim3(C), shm, st, X, d_crow + r0, (const uint16_t*)nullptr, (const uint16_t*)nullptr, m->lno_w, m->lno_b, c.norm_eps, (float*)m->q_xn.p);
            HIPCHK(hipGetLastError());
            s->prof_end(0, (double)nr * C * 8);
            s->prof_begin(nr <= 64 ? PK_GEMM_SKINNY : PK_GEMM);
            HIPCHK(cz_launch_gemm(st, (const float*)m->q_xn.p, C, m->head_n, (float*)s->logits.p, c.vocab, nullptr, 0, nr, c.vocab, C, 0));
            s->prof_end(2.0 * nr * (double)c.vocab * C, (double)c.vocab * C * 2 + (double)nr * (c.vocab + C) * 4);
            s->prof_begin(PK_CDF);
            if (p_out) HIPCHK(cz_launch_xe_probs(st, (const float*)s->logits.p, c.vocab, d_csym + r0, (double*)s->clo.p + r0, nr, c.vocab, s->fallback_dev, s->cdf_scratch, 1));
            else HIPCHK(cz_launch_cdf_pairs(st, (const float*)s->logits.p, c.vocab, d_csym + r0, (uint32_t*)s->clo.p + r0, (uint32_t*)s->chi.p + r0, nr, c.vocab, 1, s->fallback_dev, s->cdf_scratch));
            s->prof_end(0, (double)nr * c.vocab * 4);
        }
        std::vector<uint32_t> hlo(p_out ? 0 : Lr), hhi(p_out ? 0 : Lr); std::vector<double> hp(p_out ? Lr : 0);
        if (Lr && p_out) HIPCHK(hipMemcpyAsync(hp.data(), s->clo.p, Lr * 8, hipMemcpyDeviceToHost, st));
        else if (Lr) {
            HIPCHK(hipMemcpyAsync(hlo.data(), s->clo.p, Lr * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hhi.data(), s->chi.p, Lr * 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipStreamSynchronize(st));                                   // (the block's host index vectors go out of scope)
        if (p_out) for (size_t r = 0; r < Lr; ++r) p_out[cdst[r].b][cdst[r].i - first_sym[cdst[r].b]] = hp[r];
        else for (size_t r = 0; r < Lr; ++r) { lo[cdst[r].b][cdst[r].i] = hlo[r]; hi[cdst[r].b][cdst[r].i] = hhi[r]; }
    }
    return CZ_OK;
}
// every stream of the batch, groups of max_streams one after the other on this session (no lanes: the blocks are whole-device work)
cz_status cz_rwkv7_encode_seq(cz_session* s, size_t n_streams, const uint32_t* const* ids, const size_t* n, uint32_t* const* lo, uint32_t* const* hi) {
    const size_t cap = (size_t)s->max_streams;
    for (size_t g0 = 0; g0 < n_streams; g0 += cap) {
        const int B = (int)std::min(cap, n_streams - g0);
        CZCHK(rw_encode_seq_group(s, B, ids + g0, n + g0, lo + g0, hi + g0));
    }
    s->index_pos = 0;
    return CZ_OK;
}

// cross_entropy_bits_over_span_rwkv (src/main.rs:1753-1787), batched over spans: prime = history without its literal symbols ++ hint (max_context_length()
// is usize::MAX for this session, src/models.rs:150: nothing is cut); state reset; the prime fed token by token; then for every target
// p = combined_pdf_with_literals(logits)[target] (ac_p_min() past its end), bits += -log2(max(p, 1e-300)), and the model steps only on vocabulary
// targets.  On the device a span is ONE teacher-forced stream prime ++ targets through the sequence-parallel encode (the same rows, the same
// logits as the token-by-token loop: reduction orders depend on K only), evaluated from its first target on.  log2 and the running sum: host libm.
cz_status cz_rwkv7_xe_bits(cz_session* s, size_t n_spans, const uint32_t* const* hist, const size_t* n_hist, const uint32_t* const* hint, const size_t* n_hint,
                           const uint32_t* const* targets, const size_t* n_targets, double* bits_out) {
    const uint32_t V = (uint32_t)s->rwkv->cfg.vocab;
    std::vector<std::vector<uint32_t>> ids(n_spans); std::vector<std::vector<double>> p(n_spans); std::vector<std::vector<size_t>> tix(n_spans);
    std::vector<size_t> first(n_spans);
    for (size_t k = 0; k < n_spans; ++k) {
        bits_out[k] = 0.0;
        const size_t nt = n_targets[k];
        if (nt == 0) continue;                                                             // :1760
        const size_t nh = n_hist ? n_hist[k] : 0, ng = n_hint ? n_hint[k] : 0;
        for (size_t i = 0; i < nh; ++i) if (hist[k][i] < V) ids[k].push_back(hist[k][i]);  // "filter out literals from history" (:1764-1766)
        for (size_t i = 0; i < ng; ++i) {
            if (hint[k][i] >= V) { cz_set_error("span %zu: hint token %u >= vocab (the reference's embedding lookup fails there)", k, hint[k][i]); return CZ_ERR_ARG; }
            ids[k].push_back(hint[k][i]);
        }
        if (ids[k].empty()) { cz_set_error("reprime called with empty history (rwkv xe)"); return CZ_ERR_ARG; }      // :1773
        first[k] = ids[k].size() - 1;                                                      // coded symbol i = the one after ids[i]: the first target
        for (size_t i = 0; i < nt; ++i)
            if (targets[k][i] < V + 256) { ids[k].push_back(targets[k][i]); tix[k].push_back(i); }   // past the pdf's end: ac_p_min(), no step (:1779-1783)
        p[k].assign(tix[k].size(), 0.0);
    }
    const size_t cap = (size_t)s->max_streams;
    std::vector<size_t> live; for (size_t k = 0; k < n_spans; ++k) if (!tix[k].empty()) live.push_back(k);
    for (size_t g0 = 0; g0 < live.size(); g0 += cap) {
        const int B = (int)std::min(cap, live.size() - g0);
        std::vector<const uint32_t*> ip(B); std::vector<size_t> in(B), fs(B); std::vector<double*> pp(B);
        for (int b = 0; b < B; ++b) { const size_t k = live[g0 + b]; ip[b] = ids[k].data(); in[b] = ids[k].size(); fs[b] = first[k]; pp[b] = p[k].data(); }
        CZCHK(rw_encode_seq_group(s, B, ip.data(), in.data(), nullptr, nullptr, fs.data(), pp.data()));
    }
    s->index_pos = 0;
    for (size_t k = 0; k < n_spans; ++k) {
        if (n_targets[k] == 0) continue;
        std::vector<double> pt(n_targets[k], P_MIN_HOST);
        for (size_t j = 0; j < tix[k].size(); ++j) pt[tix[k][j]] = p[k][j];
        double bits = 0.0;
        for (size_t i = 0; i < n_targets[k]; ++i) { double q = pt[i]; if (!(q > 1e-300)) q = 1e-300; bits += -log2(q); }
        bits_out[k] = bits;
    }
    return CZ_OK;
}

// the decode loop in slices (begin -> advance(n)* -> fetch), like the LLaMA one: recurrent state, decoder states and ids stay in HBM
cz_status cz_rwkv7_decode_begin(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* caps, const size_t* tcs, uint32_t bos) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const uint32_t V = (uint32_t)c.vocab;
    DecRun& run = s->run;
    run.active = false;
    if (bos >= V) { cz_set_error("bos id %u >= vocab", bos); return CZ_ERR_ARG; }
    CZCHK(rw_ensure_streams(s, B));
    CZCHK(rw_reset(s));
    CZCHK(cz_dec_run_setup(s, B, payloads, lens, caps, tcs));
    cz_dec_state* dst = (cz_dec_state*)s->decst.p;
    HIPCHK(cz_launch_dec_init(s->stream, dst, B));
    hipLaunchKernelGGL(rw_init_out_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, dst, B, bos);
    hipLaunchKernelGGL(rw_fill_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, m->tok, m->mask, bos, B);
    hipLaunchKernelGGL(rw_fill_kernel, dim3((B + 63) / 64), dim3(64), 0, s->stream, m->sym, m->mask, bos, B);
    CZCHK(rw_step(s, B));                                            // step(bos) (src/main.rs:2485 equivalent for rwkv)
    const bool use_graph = !s->no_graph && !s->prof_on && cz_cdf_par_supported(c.vocab, 1);
    if (use_graph) {
        if (!m->gdec.e || m->gdec.B != B || m->gdec.gen != s->devbuf_gen)
            CZCHK(rw_graph_build(s, m->gdec, B, [&]() -> cz_status {
                CZCHK(rw_decode_iteration(s, B, 0, m->step_dev, true));   // (the step after the last symbol is harmless: its logits are never read)
                hipLaunchKernelGGL(rw_counter_kernel, dim3(1), dim3(1), 0, s->stream, m->step_dev, 1);
                return CZ_OK;
            }));
        hipLaunchKernelGGL(rw_counter_kernel, dim3(1), dim3(1), 0, s->stream, m->step_dev, 0);
    }
    run.i = 0; run.index_pos = 0; run.ec = 0; run.R = 0; run.bos = bos; run.use_graph = use_graph; run.active = true;
    return CZ_OK;
}
cz_status cz_rwkv7_decode_advance(cz_session* s, size_t n) {
    Rwkv7Model* m = s->rwkv;
    DecRun& run = s->run;
    if (!run.active) { cz_set_error("no decode in progress on this session (cz_decode_begin_batch first)"); return CZ_ERR_STATE; }
    const size_t stop = run.max_tc - run.i < n ? run.max_tc : run.i + n;
    for (size_t i = run.i; i < stop; ++i) {
        if (run.use_graph) HIPCHK(hipGraphLaunch(m->gdec.e, s->stream));
        else CZCHK(rw_decode_iteration(s, run.B, (unsigned long long)i, nullptr, i + 1 < run.max_tc));
        if (s->step_sync || ((i - run.i) % CZ_SYNC_EVERY) == CZ_SYNC_EVERY - 1) HIPCHK(hipStreamSynchronize(s->stream));   // (bounded queue depth: session.hip, decode_llama_advance)
    }
    run.i = stop;
    HIPCHK(hipStreamSynchronize(s->stream));
    return CZ_OK;
}
cz_status cz_rwkv7_decode_group(cz_session* s, int B, const uint8_t* const* payloads, const size_t* lens, const size_t* tcs, uint32_t bos,
                                uint32_t* const* out_ids) {
    CZCHK(cz_rwkv7_decode_begin(s, B, payloads, lens, nullptr, tcs, bos));
    cz_status rc = cz_rwkv7_decode_advance(s, s->run.max_tc);
    if (rc == CZ_OK) rc = cz_dec_run_fetch(s, out_ids);
    s->run.active = false;
    s->index_pos = 0;
    return rc;
}

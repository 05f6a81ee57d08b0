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

__host__ __device__ __forceinline__ int kq_idx(int k) { return (k & ~63) | ((k & 3) << 4) | ((k & 63) >> 2); }

struct RwLayer {
    // matrices, kq-packed bf16 [N][K]
    uint16_t *r, *k, *v, *o, *w1, *w2, *a1, *a2, *v1, *v2, *g1, *g2, *fk, *fv;
    // vectors (bf16)
    uint16_t *pre_w, *pre_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b, *gn_w, *gn_b, *x_r, *x_w, *x_k, *x_v, *x_a, *x_g, *w0, *a0, *v0, *k_k, *k_a,
        *r_k, *fx_k;
    int has_v;
};
struct Rwkv7Model {
    cz_rwkv7_config cfg{};
    std::vector<void*> allocs;
    uint16_t *emb = nullptr, *head = nullptr, *lno_w = nullptr, *lno_b = nullptr;
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
                                   float* __restrict__ MIX /*[6][B][C] kq*/, int B, const unsigned char* __restrict__ mask) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    float x = X[(size_t)b * C + tid];
    if (layer0) { x = block_layer_norm(x, L.pre_w, L.pre_b, eps, sh); X[(size_t)b * C + tid] = x; }
    const float xa = block_layer_norm(x, L.ln1_w, L.ln1_b, eps, sh);
    const float xx = x_prev[(size_t)b * C + tid] - xa;
    const uint16_t* mixw[6] = {L.x_r, L.x_w, L.x_k, L.x_v, L.x_a, L.x_g};
    const int kq = kq_idx(tid);
#pragma unroll
    for (int j = 0; j < 6; ++j) MIX[((size_t)j * B + b) * C + kq] = xa + xx * cz_bf16_to_f32(mixw[j][tid]);
    if (!mask || mask[b]) x_prev[(size_t)b * C + tid] = xa;                  // state.att_x_prev (:308)
}
// channel-mix prep (rwkv7.rs:472-474, 414-425)
__global__ void rw_prep_ffn_kernel(const float* __restrict__ X, const RwLayer L, float eps, float* __restrict__ x_prev, float* __restrict__ XK /*[B][C] kq*/,
                                   const unsigned char* __restrict__ mask) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const float x = X[(size_t)b * C + tid];
    const float xf = block_layer_norm(x, L.ln2_w, L.ln2_b, eps, sh);
    const float xx = x_prev[(size_t)b * C + tid] - xf;
    XK[(size_t)b * C + kq_idx(tid)] = xf + xx * cz_bf16_to_f32(L.fx_k[tid]);
    if (!mask || mask[b]) x_prev[(size_t)b * C + tid] = xf;
}
// final LayerNorm -> kq layout (rwkv7.rs:515)
__global__ void rw_ln_out_kernel(const float* __restrict__ X, const uint16_t* w, const uint16_t* bb, float eps, float* __restrict__ Y) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, tid = threadIdx.x, C = blockDim.x;
    const float y = block_layer_norm(X[(size_t)b * C + tid], w, bb, eps, sh);
    Y[(size_t)b * C + kq_idx(tid)] = y;
}
// LoRA hidden activations (rwkv7.rs:211, 229, 235, 244): tanh(w1.), a1., sigmoid(g1.), v1.  -> kq layouts padded to 64
__global__ void rw_lora_act_kernel(const float* __restrict__ P /*[B][ldp]*/, int ldp, int ow, int oa, int og, int ov, int rw, int ra, int rg, int rv,
                                   float* __restrict__ TW, float* __restrict__ TA, float* __restrict__ TG, float* __restrict__ TV, int pw, int pa, int pg, int pv) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = P + (size_t)b * ldp;
    if (t < pw) TW[(size_t)b * pw + kq_idx(t)] = t < rw ? cz_tanhf(p[ow + t]) : 0.0f;
    if (t < pa) TA[(size_t)b * pa + kq_idx(t)] = t < ra ? p[oa + t] : 0.0f;
    if (t < pg) TG[(size_t)b * pg + kq_idx(t)] = t < rg ? cz_sigmoidf(p[og + t]) : 0.0f;
    if (t < pv) TV[(size_t)b * pv + kq_idx(t)] = t < rv ? p[ov + t] : 0.0f;
}

// state kernel: block = (stream, head), 64 lanes = state rows (rwkv7.rs:213-222, 231, 239-266, 289-323, 529-542)
__global__ __launch_bounds__(64) void rw_state_kernel(const float* __restrict__ P /*[B][ldp]: r | k | v*/, int ldp, const float* __restrict__ L2 /*[B][4][C]: w a g nu*/,
                                                      const RwLayer L, int layer, int C, float* __restrict__ S /*[B][H][64][64]*/,
                                                      float* __restrict__ v_first /*[B][C]*/, float* __restrict__ OUT /*[B][C] kq*/,
                                                      const unsigned char* __restrict__ mask, int B) {
    __shared__ float kk[64], av[64], kv[64], wv[64], rv[64], vv[64], yv[64];
    const int b = blockIdx.x, h = blockIdx.y, i = threadIdx.x, c = h * 64 + i;
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
    if (layer == 0) { if (!mask || mask[b]) v_first[(size_t)b * C + c] = v; }
    else if (L.has_v) {
        const float nu = cz_sigmoidf(l2[3 * C + c] + cz_bf16_to_f32(L.v0[c]));
        const float dl = v_first[(size_t)b * C + c] - v;
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
    __syncthreads();
    kk[i] = kkn; av[i] = a; kv[i] = k; wv[i] = w; rv[i] = r; vv[i] = v;
    __syncthreads();
    float* Srow = S + (((size_t)b * gridDim.y + h) * 64 + i) * 64;
    float srow[64];
#pragma unroll
    for (int j4 = 0; j4 < 16; ++j4) {
        const float4 t4 = *reinterpret_cast<const float4*>(Srow + 4 * j4);
        srow[4 * j4] = t4.x; srow[4 * j4 + 1] = t4.y; srow[4 * j4 + 2] = t4.z; srow[4 * j4 + 3] = t4.w;
    }
    float t = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; ++j) t = cz_fma(srow[j], kk[j], t);                           // (S @ kk)[i]
    float y = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float kka = kk[j] * av[j];
        const float sn = (srow[j] * wv[j] - t * kka) + v * kv[j];
        srow[j] = sn;
        y = cz_fma(sn, rv[j], y);                                                         // (S' @ r)[i]
    }
    if (!mask || mask[b]) {
#pragma unroll
        for (int j4 = 0; j4 < 16; ++j4) *reinterpret_cast<float4*>(Srow + 4 * j4) = make_float4(srow[4 * j4], srow[4 * j4 + 1], srow[4 * j4 + 2], srow[4 * j4 + 3]);
    }
    yv[i] = y;
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
    OUT[(size_t)b * C + kq_idx(c)] = (gn + ad) * g;
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
    auto mat = [&](const std::string& name, uint16_t** out, int N, int K, float stdv) -> cz_status {
        const int Kp = pad64(K);
        CZCHK(rw_alloc(m, (void**)out, (size_t)N * Kp * 2));
        CZCHK(fetch(name, tmp, (size_t)N * K, stdv, 0.0f));
        hipLaunchKernelGGL(rw_pack_kernel, dim3(1024), dim3(256), 0, s->stream, tmp, *out, (size_t)N, K, Kp);
        HIPCHK(hipGetLastError());
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
        CZCHK(mat(p + "attn.r_proj.weight", &l.r, C, C, 0.03f)); CZCHK(mat(p + "attn.k_proj.weight", &l.k, C, C, 0.03f));
        CZCHK(mat(p + "attn.v_proj.weight", &l.v, C, C, 0.03f)); CZCHK(mat(p + "attn.o_proj.weight", &l.o, C, C, 0.03f));
        CZCHK(vec(p + "attn.g_norm.weight", &l.gn_w, C, 0.1f, 1.0f)); CZCHK(vec(p + "attn.g_norm.bias", &l.gn_b, C, 0.02f, 0.0f));
        CZCHK(vec(p + "attn.x_r", &l.x_r, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_w", &l.x_w, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_k", &l.x_k, C, 0.15f, 0.5f));
        CZCHK(vec(p + "attn.x_v", &l.x_v, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_a", &l.x_a, C, 0.15f, 0.5f)); CZCHK(vec(p + "attn.x_g", &l.x_g, C, 0.15f, 0.5f));
        CZCHK(mat(p + "attn.w_lora.lora.0.weight", &l.w1, c.decay_rank, C, 0.05f)); CZCHK(mat(p + "attn.w_lora.lora.2.weight", &l.w2, C, c.decay_rank, 0.05f));
        CZCHK(vec(p + "attn.w_lora.lora.2.bias", &l.w0, C, 1.0f, 0.0f));
        CZCHK(mat(p + "attn.a_lora.lora.0.weight", &l.a1, c.a_rank, C, 0.05f)); CZCHK(mat(p + "attn.a_lora.lora.2.weight", &l.a2, C, c.a_rank, 0.05f));
        CZCHK(vec(p + "attn.a_lora.lora.2.bias", &l.a0, C, 0.5f, 0.0f));
        if (i > 0) {                                                    // rwkv7.rs:146-152 (`contains_tensor`): the synthetic model mirrors the oracle
            l.has_v = 1;
            CZCHK(mat(p + "attn.v_lora.lora.0.weight", &l.v1, c.v_rank, C, 0.05f)); CZCHK(mat(p + "attn.v_lora.lora.2.weight", &l.v2, C, c.v_rank, 0.05f));
            CZCHK(vec(p + "attn.v_lora.lora.2.bias", &l.v0, C, 0.5f, 1.0f));
        }
        CZCHK(mat(p + "attn.g_lora.lora.0.weight", &l.g1, c.gate_rank, C, 0.05f)); CZCHK(mat(p + "attn.g_lora.lora.2.weight", &l.g2, C, c.gate_rank, 0.05f));
        CZCHK(vec(p + "attn.k_k", &l.k_k, C, 0.1f, 0.85f)); CZCHK(vec(p + "attn.k_a", &l.k_a, C, 0.1f, 1.0f)); CZCHK(vec(p + "attn.r_k", &l.r_k, C, 0.05f, 0.0f));
        CZCHK(vec(p + "ffn.x_k", &l.fx_k, C, 0.15f, 0.5f));
        CZCHK(mat(p + "ffn.key.weight", &l.fk, F, C, 0.03f)); CZCHK(mat(p + "ffn.value.weight", &l.fv, C, F, 0.03f));
    }
    CZCHK(vec("model.norm.weight", &m->lno_w, C, 0.1f, 1.0f)); CZCHK(vec("model.norm.bias", &m->lno_b, C, 0.02f, 0.0f));
    CZCHK(mat("lm_head.weight", &m->head, V, C, desc->synthetic_embed_std));
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
static cz_status rw_step(cz_session* s, int B) {
    Rwkv7Model* m = s->rwkv; const cz_rwkv7_config& c = m->cfg;
    const int C = c.hidden, F = c.d_ffn, H = C / 64, Bc = m->B;
    const int pw = pad64(c.decay_rank), pa = pad64(c.a_rank), pg = 128, pv = pad64(c.v_rank);
    const int ow = 3 * C, oa = ow + pw, og = oa + pa, ov = og + pg, ldp = ov + pv;
    hipStream_t st = s->stream;
    const size_t shm = (size_t)(C + 64) * sizeof(float);
    float *TW = m->lora1a, *TA = TW + (size_t)Bc * 64, *TG = TA + (size_t)Bc * 64, *TV = TG + (size_t)Bc * 128;
    s->prof_begin(PK_RWKV);
    HIPCHK(cz_launch_embed(st, m->tok, m->emb, m->x, B, C));
    for (int l = 0; l < c.n_layers; ++l) {
        const RwLayer& y = m->L[l];
        float* axp = m->att_x_prev + (size_t)l * Bc * C; float* fxp = m->ffn_x_prev + (size_t)l * Bc * C;
        float* S = m->att_state + (size_t)l * Bc * H * 64 * 64;
        hipLaunchKernelGGL(rw_prep_att_kernel, dim3(B), dim3(C), shm, st, m->x, y, l == 0 ? 1 : 0, c.norm_eps, axp, m->mix, Bc, m->mask);
        HIPCHK(hipGetLastError());
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
        hipLaunchKernelGGL(rw_lora_act_kernel, dim3(B), dim3(128), 0, st, m->proj, ldp, ow, oa, og, ov, c.decay_rank, c.a_rank, c.gate_rank, y.has_v ? c.v_rank : 0,
                           TW, TA, TG, TV, pw, pa, pg, pv);
        HIPCHK(hipGetLastError());
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
        hipLaunchKernelGGL(rw_state_kernel, dim3(B, H), dim3(64), 0, st, m->proj, ldp, L2, y, l, C, S, m->v_first, m->att, m->mask, Bc);
        HIPCHK(hipGetLastError());
        HIPCHK(cz_launch_dec_gemm_plain(st, m->att, C, y.o, m->x, C, m->x, B, C, C, nullptr, 0.0f));          // x = x + att
        hipLaunchKernelGGL(rw_prep_ffn_kernel, dim3(B), dim3(C), shm, st, m->x, y, c.norm_eps, fxp, m->xn, m->mask);
        HIPCHK(hipGetLastError());
        HIPCHK(cz_launch_dec_gemm_relusq(st, m->xn, C, y.fk, m->ffk, F, B, F, C));
        HIPCHK(cz_launch_dec_gemm_plain(st, m->ffk, F, y.fv, m->x, C, m->x, B, C, F, nullptr, 0.0f));          // x = x + ffn
    }
    hipLaunchKernelGGL(rw_ln_out_kernel, dim3(B), dim3(C), shm, st, m->x, m->lno_w, m->lno_b, c.norm_eps, m->xn);
    HIPCHK(hipGetLastError());
    HIPCHK(cz_launch_dec_gemm_masked(st, m->xn, C, m->head, m->logits, c.vocab, nullptr, B, c.vocab, C, m->mask));
    s->prof_end(2.0 * B * 191e6, 191e6 * 2 + (double)B * 2.36e6 * 2);
    return CZ_OK;
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
    m->cfg = pm->cfg; m->emb = pm->emb; m->head = pm->head; m->lno_w = pm->lno_w; m->lno_b = pm->lno_b; m->L = pm->L;
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

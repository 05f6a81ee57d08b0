/*
 * cz_oracle_rwkv7.c — CPU ORACLE, RWKV7 "Goose" T==1 step (test infrastructure only).
 *
 * Restates candle_rwkv7/src/models/rwkv7.rs (reference, in tree):
 *   Model::forward 509-524, Block::forward 468-478, SelfAttention::forward T==1 branch 179-328,
 *   FeedForward::forward 412-430, group_norm 529-542; session loop src/models.rs:150-160 and the RWKV
 *   branch of the token loops src/main.rs:2302-2325, 2346-2349, 2812-2834.
 * Tensor-op semantics follow candle 0.9.1 (not in tree): Linear = x . W^T, LayerNorm = fused
 * candle_nn::ops::layer_norm ((x-mean)*inv_std*w+b with var = E[x^2]-mean^2), sigmoid = 1/(exp(-x)+1),
 * mean = sum * (1/n).  Every reduction uses the canonical chunk-64 order.  "PARITY UNPINNED" vs candle.
 */
#include "cz_oracle_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define N_ 64 /* head_dim; the reference hard-wires 64e-5 = head_dim * 1e-5 for the group-norm eps (rwkv7.rs:303) */

typedef struct {
    czo_packed_w r, k, v, o, w1, w2, a1, a2, v1, v2, g1, g2, fk, fv;
    float *pre_w, *pre_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b, *gn_w, *gn_b;
    float *x_r, *x_w, *x_k, *x_v, *x_a, *x_g, *w0, *a0, *v0, *k_k, *k_a, *r_k, *fx_k;
    int has_v;
} rw_layer;

struct czo_rwkv7 {
    czo_rwkv7_config cfg;
    int n_tensors; char** names; uint16_t** tens; size_t* numel;
    rw_layer* L;
    czo_packed_w head; const uint16_t* emb; float *lno_w, *lno_b;
    /* state (rwkv7.rs:34-61) */
    float **att_x_prev, **att_state, **ffn_x_prev;
};

void czo_rwkv7_config_0b1(czo_rwkv7_config* c) { /* candle_rwkv7/convert_pth_direct.py:177-188 */
    c->vocab = 65536; c->hidden = 768; c->n_layers = 12; c->head_dim = 64; c->d_ffn = 3072;
    c->a_rank = 64; c->decay_rank = 64; c->v_rank = 32; c->gate_rank = 128; c->norm_eps = 1e-5f;
}

static const uint16_t* synth(czo_rwkv7* s, const char* name, size_t numel, uint64_t seed, float std, float mean) {
    uint16_t* t = (uint16_t*)malloc(sizeof(uint16_t) * numel);
    uint32_t tid = czo_fnv1a32(name);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < numel; ++i) t[i] = czo_synth_bf16(seed, tid, i, std, mean);
    int id = s->n_tensors++;
    s->names = (char**)realloc(s->names, sizeof(char*) * s->n_tensors);
    s->tens = (uint16_t**)realloc(s->tens, sizeof(uint16_t*) * s->n_tensors);
    s->numel = (size_t*)realloc(s->numel, sizeof(size_t) * s->n_tensors);
    s->names[id] = strdup(name); s->tens[id] = t; s->numel[id] = numel;
    return t;
}
static float* vecf(czo_rwkv7* s, const char* pfx, const char* nm, size_t n, uint64_t seed, float std, float mean) {
    char b[160]; snprintf(b, sizeof b, "%s%s", pfx, nm);
    const uint16_t* t = synth(s, b, n, seed, std, mean);
    float* f = (float*)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i) f[i] = czo_bf16_to_f32(t[i]);
    return f;
}
static void mat(czo_rwkv7* s, czo_packed_w* pw, const char* pfx, const char* nm, int N, int K, uint64_t seed, float std) {
    char b[160]; snprintf(b, sizeof b, "%s%s", pfx, nm);
    czo_pack_weights(pw, synth(s, b, (size_t)N * K, seed, std, 0.0f), N, K);
}

czo_rwkv7* czo_rwkv7_new_synthetic(const czo_rwkv7_config* cfg, uint64_t seed, float embed_std) {
    if (cfg->head_dim != N_ || cfg->hidden % N_ || cfg->vocab % 16) return NULL;
    czo_rwkv7* s = (czo_rwkv7*)calloc(1, sizeof(*s));
    s->cfg = *cfg;
    int C = cfg->hidden, L = cfg->n_layers, F = cfg->d_ffn;
    s->emb = synth(s, "model.embeddings.weight", (size_t)cfg->vocab * C, seed, embed_std, 0.0f);
    s->L = (rw_layer*)calloc(L, sizeof(rw_layer));
    for (int i = 0; i < L; ++i) {
        rw_layer* l = &s->L[i];
        char p[64]; snprintf(p, sizeof p, "model.layers.%d.", i);
        if (i == 0) { l->pre_w = vecf(s, p, "pre_norm.weight", C, seed, 0.1f, 1.0f); l->pre_b = vecf(s, p, "pre_norm.bias", C, seed, 0.02f, 0.0f); }
        l->ln1_w = vecf(s, p, "attn_norm.weight", C, seed, 0.1f, 1.0f); l->ln1_b = vecf(s, p, "attn_norm.bias", C, seed, 0.02f, 0.0f);
        l->ln2_w = vecf(s, p, "ffn_norm.weight", C, seed, 0.1f, 1.0f);  l->ln2_b = vecf(s, p, "ffn_norm.bias", C, seed, 0.02f, 0.0f);
        mat(s, &l->r, p, "attn.r_proj.weight", C, C, seed, 0.03f);
        mat(s, &l->k, p, "attn.k_proj.weight", C, C, seed, 0.03f);
        mat(s, &l->v, p, "attn.v_proj.weight", C, C, seed, 0.03f);
        mat(s, &l->o, p, "attn.o_proj.weight", C, C, seed, 0.03f);
        l->gn_w = vecf(s, p, "attn.g_norm.weight", C, seed, 0.1f, 1.0f); l->gn_b = vecf(s, p, "attn.g_norm.bias", C, seed, 0.02f, 0.0f);
        l->x_r = vecf(s, p, "attn.x_r", C, seed, 0.15f, 0.5f); l->x_w = vecf(s, p, "attn.x_w", C, seed, 0.15f, 0.5f);
        l->x_k = vecf(s, p, "attn.x_k", C, seed, 0.15f, 0.5f); l->x_v = vecf(s, p, "attn.x_v", C, seed, 0.15f, 0.5f);
        l->x_a = vecf(s, p, "attn.x_a", C, seed, 0.15f, 0.5f); l->x_g = vecf(s, p, "attn.x_g", C, seed, 0.15f, 0.5f);
        mat(s, &l->w1, p, "attn.w_lora.lora.0.weight", cfg->decay_rank, C, seed, 0.05f);
        mat(s, &l->w2, p, "attn.w_lora.lora.2.weight", C, cfg->decay_rank, seed, 0.05f);
        l->w0 = vecf(s, p, "attn.w_lora.lora.2.bias", C, seed, 1.0f, 0.0f);
        mat(s, &l->a1, p, "attn.a_lora.lora.0.weight", cfg->a_rank, C, seed, 0.05f);
        mat(s, &l->a2, p, "attn.a_lora.lora.2.weight", C, cfg->a_rank, seed, 0.05f);
        l->a0 = vecf(s, p, "attn.a_lora.lora.2.bias", C, seed, 0.5f, 0.0f);
        if (i > 0) {
            l->has_v = 1;
            mat(s, &l->v1, p, "attn.v_lora.lora.0.weight", cfg->v_rank, C, seed, 0.05f);
            mat(s, &l->v2, p, "attn.v_lora.lora.2.weight", C, cfg->v_rank, seed, 0.05f);
            l->v0 = vecf(s, p, "attn.v_lora.lora.2.bias", C, seed, 0.5f, 1.0f);
        }
        mat(s, &l->g1, p, "attn.g_lora.lora.0.weight", cfg->gate_rank, C, seed, 0.05f);
        mat(s, &l->g2, p, "attn.g_lora.lora.2.weight", C, cfg->gate_rank, seed, 0.05f);
        l->k_k = vecf(s, p, "attn.k_k", C, seed, 0.1f, 0.85f); l->k_a = vecf(s, p, "attn.k_a", C, seed, 0.1f, 1.0f);
        l->r_k = vecf(s, p, "attn.r_k", C, seed, 0.05f, 0.0f);
        l->fx_k = vecf(s, p, "ffn.x_k", C, seed, 0.15f, 0.5f);
        mat(s, &l->fk, p, "ffn.key.weight", F, C, seed, 0.03f);
        mat(s, &l->fv, p, "ffn.value.weight", C, F, seed, 0.03f);
    }
    s->lno_w = vecf(s, "", "model.norm.weight", C, seed, 0.1f, 1.0f); s->lno_b = vecf(s, "", "model.norm.bias", C, seed, 0.02f, 0.0f);
    czo_pack_weights(&s->head, synth(s, "lm_head.weight", (size_t)cfg->vocab * C, seed, embed_std, 0.0f), cfg->vocab, C);
    s->att_x_prev = calloc(L, sizeof(float*)); s->att_state = calloc(L, sizeof(float*)); s->ffn_x_prev = calloc(L, sizeof(float*));
    for (int i = 0; i < L; ++i) {
        s->att_x_prev[i] = calloc(C, sizeof(float)); s->ffn_x_prev[i] = calloc(C, sizeof(float));
        s->att_state[i] = calloc((size_t)C * N_, sizeof(float));
    }
    return s;
}
void czo_rwkv7_reset(czo_rwkv7* s) {   /* State::new, rwkv7.rs:48-61 */
    int C = s->cfg.hidden;
    for (int i = 0; i < s->cfg.n_layers; ++i) {
        memset(s->att_x_prev[i], 0, sizeof(float) * C); memset(s->ffn_x_prev[i], 0, sizeof(float) * C);
        memset(s->att_state[i], 0, sizeof(float) * (size_t)C * N_);
    }
}
void czo_rwkv7_free(czo_rwkv7* s) {
    if (!s) return;
    for (int i = 0; i < s->n_tensors; ++i) { free(s->names[i]); free(s->tens[i]); }
    free(s->names); free(s->tens); free(s->numel);
    for (int i = 0; i < s->cfg.n_layers; ++i) {
        rw_layer* l = &s->L[i];
        czo_packed_w* ws[] = {&l->r, &l->k, &l->v, &l->o, &l->w1, &l->w2, &l->a1, &l->a2, &l->v1, &l->v2, &l->g1, &l->g2, &l->fk, &l->fv};
        for (size_t j = 0; j < sizeof ws / sizeof ws[0]; ++j) free(ws[j]->p);
        float* vs[] = {l->pre_w, l->pre_b, l->ln1_w, l->ln1_b, l->ln2_w, l->ln2_b, l->gn_w, l->gn_b, l->x_r, l->x_w, l->x_k, l->x_v,
                       l->x_a, l->x_g, l->w0, l->a0, l->v0, l->k_k, l->k_a, l->r_k, l->fx_k};
        for (size_t j = 0; j < sizeof vs / sizeof vs[0]; ++j) free(vs[j]);
        free(s->att_x_prev[i]); free(s->att_state[i]); free(s->ffn_x_prev[i]);
    }
    free(s->L); free(s->head.p); free(s->lno_w); free(s->lno_b);
    free(s->att_x_prev); free(s->att_state); free(s->ffn_x_prev);
    free(s);
}
size_t czo_rwkv7_vocab(const czo_rwkv7* s) { return (size_t)s->cfg.vocab; }
const uint16_t* czo_rwkv7_tensor_bf16(const czo_rwkv7* s, const char* name, size_t* numel) {
    for (int i = 0; i < s->n_tensors; ++i)
        if (!strcmp(s->names[i], name)) { if (numel) *numel = s->numel[i]; return s->tens[i]; }
    return NULL;
}

/* candle_nn::ops::layer_norm CPU kernel: sum and sum of (rounded) squares, var = E[x^2] - mean^2 */
static void layer_norm(const float* x, const float* w, const float* b, float* y, int C, float eps) {
    float sq[64]; float s1 = 0.0f, s2 = 0.0f;
    for (int k0 = 0; k0 < C; k0 += 64) {
        int n = C - k0 < 64 ? C - k0 : 64;
        float p1 = 0.0f, p2 = 0.0f;
        for (int k = 0; k < n; ++k) sq[k] = x[k0 + k] * x[k0 + k];
        for (int k = 0; k < n; ++k) { p1 = p1 + x[k0 + k]; p2 = p2 + sq[k]; }
        s1 = k0 == 0 ? p1 : s1 + p1; s2 = k0 == 0 ? p2 : s2 + p2;
    }
    float mean = s1 / (float)C;
    float var = s2 / (float)C - mean * mean;
    float inv_std = 1.0f / sqrtf(var + eps);
    for (int k = 0; k < C; ++k) y[k] = (x[k] - mean) * inv_std * w[k] + b[k];
}
static inline float sigmoidf_(float x) { return 1.0f / (czo_expf(-x) + 1.0f); }
static void linear(const czo_packed_w* W, const float* x, float* y) { czo_gemm_canon(x, W->K, W, y, W->N, 1); }

/* SelfAttention::forward, t == 1 (rwkv7.rs:179-328) */
static void time_mix(czo_rwkv7* s, int li, const float* xs, float* out, float* v_first) {
    rw_layer* l = &s->L[li];
    int C = s->cfg.hidden, H = C / N_;
    float *xx = malloc(sizeof(float) * C * 16), *xr = xx + C, *xw = xr + C, *xk = xw + C, *xv = xk + C, *xa = xv + C, *xg = xa + C;
    float *r = xg + C, *k = r + C, *v = k + C, *w = v + C, *a = w + C, *g = a + C, *kk = g + C, *y = kk + C, *tmp = y + C;
    float lo[256];
    float* xp = s->att_x_prev[li];
    for (int c = 0; c < C; ++c) xx[c] = xp[c] - xs[c];                                   /* :193-195 */
    for (int c = 0; c < C; ++c) {                                                        /* :198-203 */
        xr[c] = xs[c] + xx[c] * l->x_r[c]; xw[c] = xs[c] + xx[c] * l->x_w[c]; xk[c] = xs[c] + xx[c] * l->x_k[c];
        xv[c] = xs[c] + xx[c] * l->x_v[c]; xa[c] = xs[c] + xx[c] * l->x_a[c]; xg[c] = xs[c] + xx[c] * l->x_g[c];
    }
    linear(&l->r, xr, r);                                                                /* :206 */
    linear(&l->w1, xw, lo); for (int j = 0; j < l->w1.N; ++j) lo[j] = czo_tanhf(lo[j]);   /* :211 */
    linear(&l->w2, lo, w);
    for (int c = 0; c < C; ++c) {                                                        /* :213-222 */
        float wc = w[c] + l->w0[c];
        float sp = czo_logf(czo_expf(-wc) + 1.0f);
        float wt = (-sp) + (-0.5f);
        w[c] = czo_expf(-czo_expf(wt));
    }
    linear(&l->k, xk, k); linear(&l->v, xv, v);                                          /* :224-225 */
    linear(&l->a1, xa, lo); linear(&l->a2, lo, a);                                       /* :228-231 */
    for (int c = 0; c < C; ++c) a[c] = sigmoidf_(a[c] + l->a0[c]);
    linear(&l->g1, xg, lo); for (int j = 0; j < l->g1.N; ++j) lo[j] = sigmoidf_(lo[j]);   /* :234-236 */
    linear(&l->g2, lo, g);
    if (li == 0) memcpy(v_first, v, sizeof(float) * C);                                  /* :239-240 */
    else if (l->has_v) {                                                                 /* :241-249 */
        linear(&l->v1, xv, lo); linear(&l->v2, lo, tmp);
        for (int c = 0; c < C; ++c) { float nu = sigmoidf_(tmp[c] + l->v0[c]); float dl = v_first[c] - v[c]; v[c] = v[c] + dl * nu; }
    }
    for (int c = 0; c < C; ++c) kk[c] = k[c] * l->k_k[c];                                /* :252 */
    for (int h = 0; h < H; ++h) {                                                        /* :253-260 */
        float sq[N_];
        for (int j = 0; j < N_; ++j) sq[j] = kk[h * N_ + j] * kk[h * N_ + j];
        float nrm = sqrtf(czo_sum(sq, N_));
        if (!(nrm > 1e-12f)) nrm = 1e-12f;
        for (int j = 0; j < N_; ++j) kk[h * N_ + j] = kk[h * N_ + j] / nrm;
    }
    for (int c = 0; c < C; ++c) { float sc = 1.0f + (a[c] - 1.0f) * l->k_a[c]; k[c] = k[c] * sc; }   /* :264-266 */
    float* S = s->att_state[li];
#pragma omp parallel for schedule(static)
    for (int hi = 0; hi < H * N_; ++hi) {                                                /* :289-304, row i of head h */
        int h = hi / N_, i = hi % N_;
        float* Si = S + (size_t)hi * N_;
        const float *kkh = kk + h * N_, *ah = a + h * N_, *kh = k + h * N_, *wh = w + h * N_, *rh = r + h * N_;
        float t = 0.0f;
        for (int j = 0; j < N_; ++j) t = __builtin_fmaf(Si[j], kkh[j], t);               /* (S @ kk)[i] */
        float vi = v[h * N_ + i];
        float yo = 0.0f;
        for (int j = 0; j < N_; ++j) {
            float kka = kkh[j] * ah[j];
            float sn = (Si[j] * wh[j] - t * kka) + vi * kh[j];
            Si[j] = sn;
            yo = __builtin_fmaf(sn, rh[j], yo);                                          /* (S' @ r)[i] */
        }
        y[hi] = yo;
    }
    memcpy(xp, xs, sizeof(float) * C);                                                   /* :308 */
    for (int h = 0; h < H; ++h) {                                                        /* group_norm :529-542, eps 64e-5 */
        float* yh = y + h * N_;
        float mean = czo_sum(yh, N_) * 0.015625f;
        float cen[N_], sq[N_];
        for (int j = 0; j < N_; ++j) { cen[j] = yh[j] - mean; sq[j] = cen[j] * cen[j]; }
        float var = czo_sum(sq, N_) * 0.015625f;
        float den = sqrtf(var + (float)64e-5);
        float rkr[N_];
        for (int j = 0; j < N_; ++j) rkr[j] = r[h * N_ + j] * k[h * N_ + j] * l->r_k[h * N_ + j];   /* :316 */
        float alpha = czo_sum(rkr, N_);
        for (int j = 0; j < N_; ++j) {
            float gn = cen[j] / den * l->gn_w[h * N_ + j] + l->gn_b[h * N_ + j];
            float ad = alpha * v[h * N_ + j];
            yh[j] = (gn + ad) * g[h * N_ + j];                                           /* :320-323 */
        }
    }
    linear(&l->o, y, out);                                                               /* :324 */
    free(xx);
}

/* FeedForward::forward (rwkv7.rs:412-430) */
static void channel_mix(czo_rwkv7* s, int li, const float* xs, float* out) {
    rw_layer* l = &s->L[li];
    int C = s->cfg.hidden, F = s->cfg.d_ffn;
    float* xk = malloc(sizeof(float) * (C + F)); float* kq = xk + C;
    float* xp = s->ffn_x_prev[li];
    for (int c = 0; c < C; ++c) { float xx = xp[c] - xs[c]; xk[c] = xs[c] + xx * l->fx_k[c]; }
    linear(&l->fk, xk, kq);
    for (int j = 0; j < F; ++j) { float z = kq[j] > 0.0f ? kq[j] : 0.0f; kq[j] = z * z; }
    linear(&l->fv, kq, out);
    memcpy(xp, xs, sizeof(float) * C);
    free(xk);
}

int czo_rwkv7_step_logits(czo_rwkv7* s, uint32_t token, float* logits) {   /* Model::forward 509-524, x[1,1] */
    int C = s->cfg.hidden;
    if (token >= (uint32_t)s->cfg.vocab) return -2;
    float *x = malloc(sizeof(float) * C * 4), *xn = x + C, *t = xn + C, *vf = t + C;
    for (int c = 0; c < C; ++c) x[c] = czo_bf16_to_f32(s->emb[(size_t)token * C + c]);
    for (int i = 0; i < s->cfg.n_layers; ++i) {                                          /* Block::forward 468-478 */
        rw_layer* l = &s->L[i];
        if (i == 0) { layer_norm(x, l->pre_w, l->pre_b, xn, C, s->cfg.norm_eps); memcpy(x, xn, sizeof(float) * C); }
        layer_norm(x, l->ln1_w, l->ln1_b, xn, C, s->cfg.norm_eps);
        time_mix(s, i, xn, t, vf);
        for (int c = 0; c < C; ++c) x[c] = x[c] + t[c];
        layer_norm(x, l->ln2_w, l->ln2_b, xn, C, s->cfg.norm_eps);
        channel_mix(s, i, xn, t);
        for (int c = 0; c < C; ++c) x[c] = x[c] + t[c];
    }
    layer_norm(x, s->lno_w, s->lno_b, xn, C, s->cfg.norm_eps);
    linear(&s->head, xn, logits);
    free(x);
    return 0;
}

/* token loops, RWKV branch (src/main.rs:1824, 1916, 2302-2325, 2346-2349; decode 2812-2834) */
size_t czo_rwkv7_encode_tokens(czo_rwkv7* s, const uint32_t* ids, size_t n, uint32_t* c_lo_out, uint32_t* c_hi_out,
                               uint8_t** payload, uint64_t* zero_width) {
    size_t V = czo_rwkv7_vocab(s);
    float* logits = malloc(sizeof(float) * V);
    uint32_t* cdf = malloc(sizeof(uint32_t) * (V + 257));
    czo_encoder e; czo_enc_init(&e);
    czo_rwkv7_reset(s);
    czo_rwkv7_step_logits(s, ids[0], logits);
    for (size_t i = 0; i + 1 < n; ++i) {
        uint32_t sym = ids[i + 1];
        czo_logits_to_cdf_rwkv(logits, V, cdf);
        if (c_lo_out) c_lo_out[i] = cdf[sym];
        if (c_hi_out) c_hi_out[i] = cdf[sym + 1];
        czo_enc_encode_counts(&e, cdf[sym], cdf[sym + 1], CZO_AC_CDF_TOTAL);
        if (sym < V) czo_rwkv7_step_logits(s, sym, logits);                              /* :2347 */
    }
    const uint8_t* out; size_t len = czo_enc_finish(&e, &out);
    if (payload) { *payload = malloc(len ? len : 1); memcpy(*payload, out, len); }
    if (zero_width) *zero_width = e.zero_width;
    czo_enc_free(&e); free(logits); free(cdf);
    return len;
}
int czo_rwkv7_decode_tokens(czo_rwkv7* s, const uint8_t* payload, size_t len, size_t token_count, uint32_t bos, uint32_t* out_ids) {
    size_t V = czo_rwkv7_vocab(s);
    float* logits = malloc(sizeof(float) * V);
    uint32_t* cdf = malloc(sizeof(uint32_t) * (V + 257));
    czo_decoder d; czo_dec_init(&d, payload, len);
    czo_rwkv7_reset(s);
    czo_rwkv7_step_logits(s, bos, logits);
    out_ids[0] = bos;
    for (size_t i = 0; i < token_count; ++i) {
        czo_logits_to_cdf_rwkv(logits, V, cdf);
        uint32_t sym = czo_dec_decode_symbol_counts(&d, cdf, V + 256, CZO_AC_CDF_TOTAL);
        out_ids[i + 1] = sym;
        if (sym < V) czo_rwkv7_step_logits(s, sym, logits);                              /* :2832 */
    }
    free(logits); free(cdf);
    return 0;
}

/* cross_entropy_bits_over_span_rwkv (src/main.rs:1753-1787), statement by statement: max_ctx = max_context_length() - 1 with
 * max_context_length() = usize::MAX for this session (src/models.rs:150), so nothing is ever cut; the history loses its literal symbols
 * (:1764-1766), the hint is appended as it is (:1767-1772); empty prime -> error (:1773); reprime = state reset + the prime token by token
 * (src/models.rs:161-178); per target: combined_pdf_with_literals, p = pdf[sym] or ac_p_min() past its end (:1777-1780),
 * bits += -log2(max(p, 1e-300)), and the model steps on vocabulary symbols only (:1782-1784).
 * Returns 0, -3 (empty prime), -4 (a prime token >= vocab: the reference's embedding lookup fails there).  p_out (optional): the n_t probabilities. */
int czo_rwkv7_xe_bits_over_span(czo_rwkv7* s, const uint32_t* history, size_t n_hist, const uint32_t* hint, size_t n_hint,
                                const uint32_t* targets, size_t n_t, double* bits_out, double* p_out) {
    *bits_out = 0.0;
    if (n_t == 0) return 0;                                              /* :1760 */
    const size_t V = czo_rwkv7_vocab(s);
    uint32_t* prime = (uint32_t*)malloc(sizeof(uint32_t) * (n_hist + n_hint + 1));
    size_t np = 0;
    for (size_t i = 0; i < n_hist; ++i) if (history[i] < V) prime[np++] = history[i];
    for (size_t i = 0; i < n_hint; ++i) prime[np++] = hint[i];
    if (np == 0) { free(prime); return -3; }
    for (size_t i = 0; i < np; ++i) if (prime[i] >= V) { free(prime); return -4; }
    float* logits = (float*)malloc(sizeof(float) * V);
    double* pdf = (double*)malloc(sizeof(double) * (V + 256));
    czo_rwkv7_reset(s);
    int rc = 0;
    for (size_t i = 0; i < np && rc == 0; ++i) rc = czo_rwkv7_step_logits(s, prime[i], logits);
    double bits = 0.0;
    for (size_t i = 0; i < n_t && rc == 0; ++i) {
        czo_combined_pdf_with_literals(logits, V, pdf);
        const size_t idx = targets[i];
        double p = idx < V + 256 ? pdf[idx] : czo_ac_p_min();
        if (p_out) p_out[i] = p;
        if (!(p > 1e-300)) p = 1e-300;
        bits += -log2(p);
        if (idx < V) rc = czo_rwkv7_step_logits(s, targets[i], logits);
    }
    free(prime); free(logits); free(pdf);
    *bits_out = bits;
    return rc;
}

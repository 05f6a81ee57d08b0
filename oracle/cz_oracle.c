/*
 * cz_oracle.c — CPU ORACLE (test infrastructure only; see cz_oracle.h for the parity statement).
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root,
 * turtle261/candlezip @ 2025-10-10).  Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma
 * -ffp-contract=off -fopenmp).  -ffp-contract=off is REQUIRED: every fused multiply-add below is
 * written explicitly as __builtin_fmaf; nothing else may be contracted.
 */
#include "cz_oracle_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

int czo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void czo_free(void* p) { free(p); }
void czo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ======================================================================================
 * Scalar numerics
 * ====================================================================================== */

#define bits_f32 czo_bits_f32
#define f32_bits czo_f32_bits
#define bf16_to_f32 czo_bf16_to_f32
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = f32_bits(f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

/* Owned expf: the reference calls f32::exp (src/main.rs:790, candle softmax/silu), i.e.
 * the host libm, which is not reproducible across hosts.  The oracle and the HIP kernels share this
 * definition instead: Cody-Waite reduction by ln2 (hi/lo), degree-5 Cephes polynomial, two-step
 * power-of-two scaling so that subnormal results are rounded exactly once.  Only IEEE +,*,fma. */
float czo_expf(float x) {
    if (x != x) return x;
    if (x < -104.0f) x = -104.0f;
    if (x > 89.0f) x = 89.0f;
    const float MAGIC = 12582912.0f; /* 1.5 * 2^23 */
    float t = __builtin_fmaf(x, 1.44269504088896341f, MAGIC);
    float n = t - MAGIC;
    int ni = (int)n;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);        /* ln2 hi (0x3f317200) */
    r = __builtin_fmaf(n, -1.42860682030941723e-06f, r);         /* ln2 lo */
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    int k1 = ni / 2, k2 = ni - k1;                               /* both in [-75, 64] */
    float s1 = bits_f32((uint32_t)(k1 + 127) << 23);
    float s2 = bits_f32((uint32_t)(k2 + 127) << 23);
    return (y * s1) * s2;
}

/* Owned logf for x > 0 (RWKV7 softplus, rwkv7.rs:206-209).  x = m * 2^e, m in [sqrt(.5), sqrt(2));
 * log(m) by the Cephes logf polynomial. */
float czo_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return bits_f32(0x7fc00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t u = f32_bits(x);
    int e = 0;
    if ((u >> 23) == 0) { x = x * 8388608.0f; u = f32_bits(x); e = -23; }
    e += (int)(u >> 23) - 126;
    float m = bits_f32((u & 0x007fffffu) | 0x3f000000u);        /* [0.5, 1) */
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float res = m + y;
    res = __builtin_fmaf(fe, 0.693359375f, res);
    return res;
}

/* Owned tanhf (RWKV7 w-lora, rwkv7.rs:200): odd; |x| < 0.625 by the Cephes polynomial, else via expf. */
float czo_tanhf(float x) {
    if (x != x) return x;
    float ax = x < 0.0f ? -x : x;
    if (ax >= 9.0f) return x < 0.0f ? -1.0f : 1.0f;
    if (ax >= 0.625f) {
        float e = czo_expf(ax + ax);
        float r = 1.0f - 2.0f / (e + 1.0f);
        return x < 0.0f ? -r : r;
    }
    float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    p = p * z;
    return __builtin_fmaf(p, x, x);
}

/* canonical reductions (see header) */
float czo_dot(const float* a, const float* b, int K) {
    float t = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        int ke = k0 + 64 < K ? k0 + 64 : K;
        float p = 0.0f;
        for (int k = k0; k < ke; ++k) p = __builtin_fmaf(a[k], b[k], p);
        t = (k0 == 0) ? p : t + p;
    }
    return t;
}
float czo_sum(const float* a, int K) {
    float t = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        int ke = k0 + 64 < K ? k0 + 64 : K;
        float p = 0.0f;
        for (int k = k0; k < ke; ++k) p = p + a[k];
        t = (k0 == 0) ? p : t + p;
    }
    return t;
}

/* ======================================================================================
 * A-2 / A-3 / R-2 : logits -> pdf -> cdf
 * ====================================================================================== */

double czo_ac_p_min(void) { return 2.0 * pow(2.0, -30.0); } /* src/main.rs:235-238: 2*2^-(32-2) */

/* src/main.rs:784-801.  max in f32; (l - max) and exp in f32; widen; sequential f64 sum; divide. */
void czo_softmax_pdf(const float* logits, size_t vocab, double* pdf) {
    float max = -INFINITY;
    for (size_t i = 0; i < vocab; ++i) if (logits[i] > max) max = logits[i];
    double sum = 0.0;
    for (size_t i = 0; i < vocab; ++i) {
        double e = (double)czo_expf(logits[i] - max);
        pdf[i] = e; sum += e;
    }
    if (sum <= 0.0) {                                   /* :794-798 */
        double u = 1.0 / (double)vocab;
        for (size_t i = 0; i < vocab; ++i) pdf[i] = u;
        return;
    }
    for (size_t i = 0; i < vocab; ++i) pdf[i] = pdf[i] / sum;
}

/* src/main.rs:758-767 */
void czo_softmax_pdf_floor(const float* logits, size_t vocab, double p_floor, double* pdf) {
    float max = -INFINITY;
    for (size_t i = 0; i < vocab; ++i) if (logits[i] > max) max = logits[i];
    double sum = 0.0;
    for (size_t i = 0; i < vocab; ++i) {
        double e = (double)czo_expf(logits[i] - max);
        pdf[i] = e; sum += e;
    }
    for (size_t i = 0; i < vocab; ++i) { double p = pdf[i] / sum; pdf[i] = p > p_floor ? p : p_floor; } /* f64::max */
    double norm = 0.0;
    for (size_t i = 0; i < vocab; ++i) norm += pdf[i];
    for (size_t i = 0; i < vocab; ++i) pdf[i] /= norm;
}

/* src/main.rs:769-782 */
void czo_combined_pdf_with_literals(const float* logits, size_t vocab, double* pdf) {
    czo_softmax_pdf_floor(logits, vocab, czo_ac_p_min(), pdf);
    double p_escape_total = 256.0 * czo_ac_p_min();
    double scale = 1.0 - p_escape_total; if (!(scale > 0.0)) scale = 0.0;
    for (size_t i = 0; i < vocab; ++i) pdf[i] *= scale;
    double p_literal_each = p_escape_total > 0.0 ? p_escape_total / 256.0 : 0.0;
    for (size_t i = 0; i < 256; ++i) pdf[vocab + i] = p_literal_each;
    double sum = 0.0;
    for (size_t i = 0; i < vocab + 256; ++i) sum += pdf[i];
    if (sum > 0.0) for (size_t i = 0; i < vocab + 256; ++i) pdf[i] /= sum;
}

/* src/main.rs:805-824 (and the identical inline quantiser at :2305-2321 / :2815-2827) */
void czo_quantize_pdf_to_cdf(const double* pdf, size_t n, uint32_t* cdf) {
    cdf[0] = 0;
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        acc += pdf[i];
        double f = floor(acc * (double)CZO_AC_CDF_TOTAL);
        int64_t v;                                     /* Rust `as i64` saturates; NaN -> 0 */
        if (f != f) v = 0;
        else if (f >= 9.2233720368547758e18) v = INT64_MAX;
        else if (f <= -9.2233720368547758e18) v = INT64_MIN;
        else v = (int64_t)f;
        if (v < 0) v = 0;
        if (v > (int64_t)CZO_AC_CDF_TOTAL) v = (int64_t)CZO_AC_CDF_TOTAL;
        int64_t prev = (int64_t)cdf[i];
        if (v < prev) v = prev;
        cdf[i + 1] = (uint32_t)v;
    }
    cdf[n] = CZO_AC_CDF_TOTAL;
}

void czo_logits_to_cdf_smollm(const float* logits, size_t vocab, uint32_t* cdf) {
    double* pdf = (double*)malloc(sizeof(double) * vocab);
    czo_softmax_pdf(logits, vocab, pdf);
    czo_quantize_pdf_to_cdf(pdf, vocab, cdf);
    free(pdf);
}
void czo_logits_to_cdf_rwkv(const float* logits, size_t vocab, uint32_t* cdf) {
    double* pdf = (double*)malloc(sizeof(double) * (vocab + 256));
    czo_combined_pdf_with_literals(logits, vocab, pdf);
    czo_quantize_pdf_to_cdf(pdf, vocab + 256, cdf);
    free(pdf);
}

/* ======================================================================================
 * A-4 : arithmetic encoder  (src/main.rs:261-318, 353-404)
 * ====================================================================================== */
#define AC_FULL 0xFFFFFFFFull          /* mask   = 2^32 - 1   (:277) */
#define AC_HALF 0x80000000ull          /* b_to_pm1 = 2^31     (:279) */
#define AC_QUARTER 0x40000000ull       /* b_to_pm2 = 2^30     (:280) */

void czo_enc_init(czo_encoder* e) {
    memset(e, 0, sizeof(*e));
    e->low = 0; e->high = AC_FULL;
}
static void enc_write_byte(czo_encoder* e, uint8_t b) {
    if (e->out_len == e->out_cap) {
        e->out_cap = e->out_cap ? e->out_cap * 2 : 4096;
        e->out = (uint8_t*)realloc(e->out, e->out_cap);
    }
    e->out[e->out_len++] = b;
}
static void enc_put_bit_internal(czo_encoder* e, uint8_t bit) {      /* :298-308 */
    e->bit_buffer = (uint8_t)((e->bit_buffer << 1) | (bit & 1));
    e->bit_count += 1;
    if (e->bit_count == 8) { enc_write_byte(e, e->bit_buffer); e->bit_buffer = 0; e->bit_count = 0; }
}
static void enc_put_bit(czo_encoder* e, uint8_t bit) {               /* :310-317 */
    enc_put_bit_internal(e, bit);
    while (e->carry_run > 0) { enc_put_bit_internal(e, (uint8_t)((~bit) & 1)); e->carry_run -= 1; }
}
void czo_enc_encode_counts(czo_encoder* e, uint64_t c_lo, uint64_t c_hi, uint64_t total) { /* :353-385 */
    if (c_lo == c_hi) e->zero_width += 1;                             /* H3 diagnostics only */
    u128 range = (u128)(uint64_t)(e->high - e->low + 1);              /* u64 arithmetic, then widened */
    u128 new_low = (u128)e->low + (range * (u128)c_lo) / (u128)total;
    u128 new_high = (u128)e->low + (range * (u128)c_hi) / (u128)total - 1;   /* wraps in u128 if 0 (release build) */
    e->low = (uint64_t)(new_low & (u128)AC_FULL);
    e->high = (uint64_t)(new_high & (u128)AC_FULL);
    for (;;) {
        if (e->high < AC_HALF) {
            enc_put_bit(e, 0);
        } else if (e->low >= AC_HALF) {
            enc_put_bit(e, 1);
            e->low -= AC_HALF; e->high -= AC_HALF;
        } else if (e->low >= AC_QUARTER && e->high < AC_QUARTER * 3) {
            e->carry_run += 1;
            e->low -= AC_QUARTER; e->high -= AC_QUARTER;
        } else break;
        e->low = (e->low << 1) & AC_FULL;
        e->high = ((e->high << 1) & AC_FULL) | 1;
    }
}
size_t czo_enc_finish(czo_encoder* e, const uint8_t** out) {          /* :387-399 */
    e->carry_run += 1;
    if (e->low < AC_QUARTER) enc_put_bit(e, 0); else enc_put_bit(e, 1);
    if (e->bit_count > 0) {
        int remaining = 8 - e->bit_count;
        for (int i = 0; i < remaining; ++i) enc_put_bit_internal(e, 0);
    }
    if (out) *out = e->out;
    return e->out_len;
}
void czo_enc_free(czo_encoder* e) { free(e->out); e->out = NULL; e->out_len = e->out_cap = 0; }

/* ======================================================================================
 * A-5 : arithmetic decoder  (src/main.rs:406-449, 500-548)
 * ====================================================================================== */
static int dec_get_bit(czo_decoder* d) {                              /* :438-449; -1 = None */
    if (d->byte_pos >= d->in_len) return -1;
    uint8_t byte = d->in[d->byte_pos];
    int bit = (byte >> (7 - d->bit_pos)) & 1;
    d->bit_pos += 1;
    if (d->bit_pos >= 8) { d->bit_pos = 0; d->byte_pos += 1; }
    return bit;
}
void czo_dec_init(czo_decoder* d, const uint8_t* in, size_t len) {   /* :419-436 */
    d->low = 0; d->high = AC_FULL; d->code = 0;
    d->in = in; d->in_len = len; d->byte_pos = 0; d->bit_pos = 0;
    for (int i = 0; i < 32; ++i) { int b = dec_get_bit(d); d->code = (d->code << 1) | (uint64_t)(b < 0 ? 1 : b); }
}
uint32_t czo_dec_decode_symbol_counts(czo_decoder* d, const uint32_t* cdf, size_t n, uint32_t total) { /* :500-548 */
    uint64_t range64 = d->high - d->low + 1;
    u128 value = (((u128)(uint64_t)(d->code - d->low + 1) * (u128)total) - 1) / (u128)range64;
    uint32_t value_u = (uint32_t)value;
    size_t lo = 0, hi = n;                                            /* cdf.len() - 1 */
    while (lo + 1 < hi) { size_t mid = (lo + hi) / 2; if (cdf[mid] <= value_u) lo = mid; else hi = mid; }
    size_t s = lo;
    uint64_t c_lo = cdf[s], c_hi = cdf[s + 1];
    u128 range = (u128)range64;
    u128 new_low = (u128)d->low + (range * (u128)c_lo) / (u128)total;
    u128 new_high = (u128)d->low + (range * (u128)c_hi) / (u128)total - 1;
    d->low = (uint64_t)new_low;                                       /* NOT masked (unlike the encoder) */
    d->high = (uint64_t)new_high;
    for (;;) {
        if (d->high < AC_HALF) {
        } else if (d->low >= AC_HALF) {
            d->low -= AC_HALF; d->high -= AC_HALF; d->code -= AC_HALF;
        } else if (d->low >= AC_QUARTER && d->high < AC_QUARTER * 3) {
            d->low -= AC_QUARTER; d->high -= AC_QUARTER; d->code -= AC_QUARTER;
        } else break;
        d->low = (d->low << 1) & AC_FULL;
        d->high = ((d->high << 1) & AC_FULL) | 1;
        int b = dec_get_bit(d);
        d->code = ((d->code << 1) & AC_FULL) | (uint64_t)(b < 0 ? 1 : b);
    }
    return (uint32_t)s;
}

/* ======================================================================================
 * A-8 : container header v2  (src/main.rs:551-646)
 * ====================================================================================== */
size_t czo_write_var_u64(uint64_t v, uint8_t* out) {                  /* :566-575 */
    size_t n = 0;
    while (v >= 0x80) { out[n++] = (uint8_t)((v & 0x7F) | 0x80); v >>= 7; }
    out[n++] = (uint8_t)v;
    return n;
}
size_t czo_read_var_u64(const uint8_t* in, size_t len, uint64_t* v) { /* :577-592 */
    uint32_t shift = 0; uint64_t out = 0; size_t n = 0;
    for (;;) {
        if (n >= len) return 0;
        uint8_t byte = in[n++];
        out |= (uint64_t)(byte & 0x7F) << shift;
        if ((byte & 0x80) == 0) break;
        shift += 7;
        if (shift > 63) return 0;
    }
    *v = out; return n;
}
uint32_t czo_flags_pack(int agent_used, int agent_mock, int gates_present, uint32_t agent_chunk) { /* :247-255 */
    uint32_t f = 0;
    if (agent_used) f |= 1u << 0;
    if (agent_mock) f |= 1u << 1;
    if (gates_present) f |= 1u << 2;
    f |= (agent_chunk & 0xFFFFu) << 16;
    return f;
}
static void put_u32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

size_t czo_write_header_v2(const czo_header* h, const uint8_t* repr, uint8_t* out, size_t cap) { /* :594-611 */
    if (cap < 4 + 2 + 4 + 10 + 10 + 48 + 20 + (size_t)h->model_file_repr_len) return 0;
    size_t n = 0;
    put_u32(out + n, CZO_MAGIC); n += 4;
    out[n++] = (uint8_t)(CZO_VERSION & 0xFF); out[n++] = (uint8_t)(CZO_VERSION >> 8);
    put_u32(out + n, h->bos_token_id); n += 4;
    n += czo_write_var_u64(h->token_count, out + n);
    n += czo_write_var_u64(h->orig_len_bytes, out + n);
    memcpy(out + n, h->model_hash16, 16); n += 16;
    memcpy(out + n, h->tokenizer_hash16, 16); n += 16;
    memcpy(out + n, h->orig_hash16, 16); n += 16;
    put_u32(out + n, h->reserved_flags); n += 4;
    put_u32(out + n, h->context_window); n += 4;
    put_u32(out + n, h->vocab_size); n += 4;
    put_u32(out + n, h->model_file_repr_len); n += 4;
    put_u32(out + n, h->reprime_interval); n += 4;
    memcpy(out + n, repr, h->model_file_repr_len); n += h->model_file_repr_len;
    return n;
}
size_t czo_read_header_v2(const uint8_t* buf, size_t len, czo_header* h, size_t* repr_off) { /* :613-646 */
    size_t n = 0, k;
    if (len < 10) return 0;
    if (get_u32(buf) != CZO_MAGIC) return 0;
    n = 4;
    if ((uint16_t)(buf[n] | (buf[n + 1] << 8)) != CZO_VERSION) return 0;
    n += 2;
    h->bos_token_id = get_u32(buf + n); n += 4;
    if (!(k = czo_read_var_u64(buf + n, len - n, &h->token_count))) return 0;
    n += k;
    if (!(k = czo_read_var_u64(buf + n, len - n, &h->orig_len_bytes))) return 0;
    n += k;
    if (len - n < 48 + 20) return 0;
    memcpy(h->model_hash16, buf + n, 16); n += 16;
    memcpy(h->tokenizer_hash16, buf + n, 16); n += 16;
    memcpy(h->orig_hash16, buf + n, 16); n += 16;
    h->reserved_flags = get_u32(buf + n); n += 4;
    h->context_window = get_u32(buf + n); n += 4;
    h->vocab_size = get_u32(buf + n); n += 4;
    h->model_file_repr_len = get_u32(buf + n); n += 4;
    h->reprime_interval = get_u32(buf + n); n += 4;
    if (len - n < h->model_file_repr_len) return 0;
    if (repr_off) *repr_off = n;
    n += h->model_file_repr_len;
    return n;
}

/* ======================================================================================
 * Synthetic weights (no checkpoints exist offline).  Counter-based, integer-only up to
 * one f32 multiply + add, so that any host reproduces the same bf16 bits.
 * ====================================================================================== */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint32_t czo_fnv1a32(const char* s) {
    uint32_t h = 2166136261u;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 16777619u; }
    return h;
}
uint16_t czo_synth_bf16(uint64_t seed, uint32_t tid, uint64_t idx, float std, float mean) {
    uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)tid << 32) ^ idx));
    int s = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48) - 131070;
    float scale = std * 2.6428893e-05f;          /* 1 / (65536/sqrt(3)) : Irwin-Hall(4) -> unit variance */
    float v = (float)s * scale;
    v = v + mean;
    return f32_to_bf16_rne(v);
}

/* ======================================================================================
 * Canonical GEMM   Y[m][n] = cz_dot(X[m][:], W[n][:])   (weights given as bf16 [N][K])
 * Packed for the CPU as f32 panels of 16 output columns: P[n/16][k][16].
 * ====================================================================================== */
#define NR 16
typedef czo_packed_w packed_w;

void czo_pack_weights(packed_w* pw, const uint16_t* w_bf16, int N, int K) {
    pw->N = N; pw->K = K;
    int np = (N + NR - 1) / NR;
    pw->p = (float*)aligned_alloc(64, sizeof(float) * (size_t)np * K * NR);
#pragma omp parallel for schedule(static)
    for (int pi = 0; pi < np; ++pi)
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < NR; ++j) {
                int n = pi * NR + j;
                pw->p[((size_t)pi * K + k) * NR + j] = n < N ? bf16_to_f32(w_bf16[(size_t)n * K + k]) : 0.0f;
            }
}

#define GEMM_KERNEL(MR)                                                                               \
    static inline void gemm_kernel_##MR(const float* X, int ldx, const float* P, int K, float* Y, int ldy, int ncols) { \
        float tot[MR][NR];                                                                            \
        for (int k0 = 0; k0 < K; k0 += 64) {                                                          \
            int ke = k0 + 64 < K ? k0 + 64 : K;                                                       \
            float p[MR][NR];                                                                          \
            for (int r = 0; r < MR; ++r) for (int j = 0; j < NR; ++j) p[r][j] = 0.0f;                 \
            for (int k = k0; k < ke; ++k) {                                                           \
                const float* w = P + (size_t)k * NR;                                                  \
                for (int r = 0; r < MR; ++r) {                                                        \
                    float xv = X[(size_t)r * ldx + k];                                                \
                    for (int j = 0; j < NR; ++j) p[r][j] = __builtin_fmaf(xv, w[j], p[r][j]);         \
                }                                                                                     \
            }                                                                                         \
            if (k0 == 0) { for (int r = 0; r < MR; ++r) for (int j = 0; j < NR; ++j) tot[r][j] = p[r][j]; } \
            else { for (int r = 0; r < MR; ++r) for (int j = 0; j < NR; ++j) tot[r][j] = tot[r][j] + p[r][j]; } \
        }                                                                                             \
        for (int r = 0; r < MR; ++r) for (int j = 0; j < ncols; ++j) Y[(size_t)r * ldy + j] = tot[r][j]; \
    }
GEMM_KERNEL(1)
GEMM_KERNEL(2)
GEMM_KERNEL(4)

void czo_gemm_canon(const float* X, int ldx, const packed_w* W, float* Y, int ldy, int M) {
    int np = (W->N + NR - 1) / NR, K = W->K;
    int mb = (M + 3) / 4;
#pragma omp parallel for collapse(2) schedule(static)
    for (int pi = 0; pi < np; ++pi)
        for (int bi = 0; bi < mb; ++bi) {
            int m0 = bi * 4, mr = M - m0 < 4 ? M - m0 : 4;
            const float* P = W->p + (size_t)pi * K * NR;
            int ncols = W->N - pi * NR < NR ? W->N - pi * NR : NR;
            const float* x = X + (size_t)m0 * ldx;
            float* y = Y + (size_t)m0 * ldy + (size_t)pi * NR;
            if (mr == 4) gemm_kernel_4(x, ldx, P, K, y, ldy, ncols);
            else {
                int r = 0;
                if (mr - r >= 2) { gemm_kernel_2(x, ldx, P, K, y, ldy, ncols); r += 2; }
                if (mr - r >= 1) gemm_kernel_1(x + (size_t)r * ldx, ldx, P, K, y + (size_t)r * ldy, ldy, ncols);
            }
        }
}

/* ======================================================================================
 * A-1 / A-1b : SmolLM2 / LLaMA forward.
 * Follows candle-transformers 0.9.1 models::llama (NOT in tree; call sites src/models.rs:53-58, 94,
 * 110) as recalled in SURVEY.md §8(c): pre-norm residual blocks, GQA, non-interleaved RoPE, f32
 * attention with scores / sqrt(head_dim), softmax_last_dim, SwiGLU, final RMSNorm, tied lm_head on
 * the LAST row only.  Reduction order = canonical (header).  "PARITY UNPINNED" vs candle.
 * ====================================================================================== */
#define HD 64
struct czo_smollm {
    czo_llama_config cfg;
    /* bf16 originals (exported to tests) */
    int n_tensors; char** names; uint16_t** tens; size_t* numel;
    /* packed f32 */
    packed_w *wq, *wk, *wv, *wo, *wg, *wu, *wd; packed_w head;
    float **ln1, **ln2, *lnf;
    float *rope_cos, *rope_sin;          /* [max_pos][32] */
    /* state */
    size_t index_pos, cache_cap;
    float **kc, **vc;                    /* per layer [cache_cap][n_kv*64] */
};

void czo_llama_config_smollm2_135m(czo_llama_config* c) {
    c->vocab = 49152; c->d_model = 576; c->n_layers = 30; c->n_heads = 9; c->n_kv_heads = 3;
    c->d_ffn = 1536; c->max_pos = 8192; c->rope_theta = 100000.0f; c->rms_eps = 1e-5f;
}

static uint16_t* synth_tensor(czo_smollm* s, const char* name, size_t numel, uint64_t seed, float std, float mean) {
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
static float* widen(const uint16_t* t, size_t n) {
    float* f = (float*)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i) f[i] = bf16_to_f32(t[i]);
    return f;
}

/* RoPE tables: candle llama Cache::new — inv_freq_i = 1/theta^(i/head_dim) (f32), angle = pos*inv_freq
 * (f32 product), cos/sin in f32.  The libm f32 cos/sin/powf of the reference host are not reproducible
 * (H5); we round the f64 libm result to f32 (== correctly rounded f32 in all but ~1e-9 of cases). */
static void build_rope(czo_smollm* s) {
    int mp = s->cfg.max_pos;
    s->rope_cos = (float*)malloc(sizeof(float) * (size_t)mp * 32);
    s->rope_sin = (float*)malloc(sizeof(float) * (size_t)mp * 32);
    float inv[32];
    for (int i = 0; i < 32; ++i) {
        float e = (float)(2 * i) / (float)HD;
        float pw = (float)pow((double)s->cfg.rope_theta, (double)e);
        inv[i] = 1.0f / pw;
    }
    for (int p = 0; p < mp; ++p)
        for (int i = 0; i < 32; ++i) {
            float a = (float)p * inv[i];
            s->rope_cos[(size_t)p * 32 + i] = (float)cos((double)a);
            s->rope_sin[(size_t)p * 32 + i] = (float)sin((double)a);
        }
}

czo_smollm* czo_smollm_new_synthetic(const czo_llama_config* cfg, uint64_t seed, float embed_std) {
    czo_smollm* s = (czo_smollm*)calloc(1, sizeof(*s));
    s->cfg = *cfg;
    int L = cfg->n_layers, d = cfg->d_model, kv = cfg->n_kv_heads * HD, f = cfg->d_ffn;
    if (cfg->n_heads * HD != d || d % 64 || f % 64 || cfg->vocab % 16) { free(s); return NULL; }
    char nm[128];
    const uint16_t* emb = synth_tensor(s, "model.embed_tokens.weight", (size_t)cfg->vocab * d, seed, embed_std, 0.0f);
    czo_pack_weights(&s->head, emb, cfg->vocab, d);            /* tie_word_embeddings = true */
    s->wq = calloc(L, sizeof(packed_w)); s->wk = calloc(L, sizeof(packed_w)); s->wv = calloc(L, sizeof(packed_w));
    s->wo = calloc(L, sizeof(packed_w)); s->wg = calloc(L, sizeof(packed_w)); s->wu = calloc(L, sizeof(packed_w));
    s->wd = calloc(L, sizeof(packed_w));
    s->ln1 = calloc(L, sizeof(float*)); s->ln2 = calloc(L, sizeof(float*));
    for (int l = 0; l < L; ++l) {
        snprintf(nm, sizeof nm, "model.layers.%d.input_layernorm.weight", l);
        s->ln1[l] = widen(synth_tensor(s, nm, d, seed, 0.1f, 1.0f), d);
        snprintf(nm, sizeof nm, "model.layers.%d.self_attn.q_proj.weight", l);
        czo_pack_weights(&s->wq[l], synth_tensor(s, nm, (size_t)d * d, seed, 0.02f, 0.0f), d, d);
        snprintf(nm, sizeof nm, "model.layers.%d.self_attn.k_proj.weight", l);
        czo_pack_weights(&s->wk[l], synth_tensor(s, nm, (size_t)kv * d, seed, 0.02f, 0.0f), kv, d);
        snprintf(nm, sizeof nm, "model.layers.%d.self_attn.v_proj.weight", l);
        czo_pack_weights(&s->wv[l], synth_tensor(s, nm, (size_t)kv * d, seed, 0.02f, 0.0f), kv, d);
        snprintf(nm, sizeof nm, "model.layers.%d.self_attn.o_proj.weight", l);
        czo_pack_weights(&s->wo[l], synth_tensor(s, nm, (size_t)d * d, seed, 0.02f, 0.0f), d, d);
        snprintf(nm, sizeof nm, "model.layers.%d.post_attention_layernorm.weight", l);
        s->ln2[l] = widen(synth_tensor(s, nm, d, seed, 0.1f, 1.0f), d);
        snprintf(nm, sizeof nm, "model.layers.%d.mlp.gate_proj.weight", l);
        czo_pack_weights(&s->wg[l], synth_tensor(s, nm, (size_t)f * d, seed, 0.02f, 0.0f), f, d);
        snprintf(nm, sizeof nm, "model.layers.%d.mlp.up_proj.weight", l);
        czo_pack_weights(&s->wu[l], synth_tensor(s, nm, (size_t)f * d, seed, 0.02f, 0.0f), f, d);
        snprintf(nm, sizeof nm, "model.layers.%d.mlp.down_proj.weight", l);
        czo_pack_weights(&s->wd[l], synth_tensor(s, nm, (size_t)d * f, seed, 0.02f, 0.0f), d, f);
    }
    s->lnf = widen(synth_tensor(s, "model.norm.weight", d, seed, 0.1f, 1.0f), d);
    build_rope(s);
    s->cache_cap = 0; s->kc = calloc(L, sizeof(float*)); s->vc = calloc(L, sizeof(float*));
    s->index_pos = 0;
    return s;
}

void czo_smollm_free(czo_smollm* s) {
    if (!s) return;
    int L = s->cfg.n_layers;
    for (int i = 0; i < s->n_tensors; ++i) { free(s->names[i]); free(s->tens[i]); }
    free(s->names); free(s->tens); free(s->numel);
    for (int l = 0; l < L; ++l) {
        free(s->wq[l].p); free(s->wk[l].p); free(s->wv[l].p); free(s->wo[l].p);
        free(s->wg[l].p); free(s->wu[l].p); free(s->wd[l].p);
        free(s->ln1[l]); free(s->ln2[l]); free(s->kc[l]); free(s->vc[l]);
    }
    free(s->wq); free(s->wk); free(s->wv); free(s->wo); free(s->wg); free(s->wu); free(s->wd);
    free(s->ln1); free(s->ln2); free(s->lnf); free(s->head.p);
    free(s->rope_cos); free(s->rope_sin); free(s->kc); free(s->vc);
    free(s);
}
size_t czo_smollm_vocab(const czo_smollm* s) { return (size_t)s->cfg.vocab; }
size_t czo_smollm_index_pos(const czo_smollm* s) { return s->index_pos; }
const uint16_t* czo_smollm_tensor_bf16(const czo_smollm* s, const char* name, size_t* numel) {
    for (int i = 0; i < s->n_tensors; ++i)
        if (!strcmp(s->names[i], name)) { if (numel) *numel = s->numel[i]; return s->tens[i]; }
    return NULL;
}

static void ensure_cache(czo_smollm* s, size_t need) {
    if (need <= s->cache_cap) return;
    size_t cap = s->cache_cap ? s->cache_cap : 1024;
    while (cap < need) cap *= 2;
    size_t row = (size_t)s->cfg.n_kv_heads * HD;
    for (int l = 0; l < s->cfg.n_layers; ++l) {
        s->kc[l] = (float*)realloc(s->kc[l], sizeof(float) * cap * row);
        s->vc[l] = (float*)realloc(s->vc[l], sizeof(float) * cap * row);
    }
    s->cache_cap = cap;
}

/* candle_nn::ops::rms_norm (CPU): m = sqrt(sum(x^2)/d + eps);  y = x / m * w   (x*x rounded, then summed) */
static void rms_norm_rows(const float* X, const float* w, float* Y, int M, int d, float eps) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* x = X + (size_t)m * d; float* y = Y + (size_t)m * d;
        float sq[64]; float t = 0.0f;
        for (int k0 = 0; k0 < d; k0 += 64) {
            int n = d - k0 < 64 ? d - k0 : 64;
            for (int k = 0; k < n; ++k) sq[k] = x[k0 + k] * x[k0 + k];
            float p = 0.0f; for (int k = 0; k < n; ++k) p = p + sq[k];
            t = (k0 == 0) ? p : t + p;
        }
        float mm = sqrtf(t / (float)d + eps);
        for (int k = 0; k < d; ++k) y[k] = x[k] / mm * w[k];
    }
}

/* candle_nn::rotary_emb::rope (non-interleaved halves): products rounded separately, no fma */
static void rope_rows(float* X, int M, int nh, size_t pos0, const float* cs, const float* sn) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* c = cs + (pos0 + m) * 32; const float* s = sn + (pos0 + m) * 32;
        for (int h = 0; h < nh; ++h) {
            float* x = X + ((size_t)m * nh + h) * HD;
            for (int i = 0; i < 32; ++i) {
                float x1 = x[i], x2 = x[i + 32];
                float a = x1 * c[i], b = x2 * s[i], cc = x1 * s[i], dd = x2 * c[i];
                x[i] = a - b; x[i + 32] = cc + dd;
            }
        }
    }
}

/* attention for rows at absolute positions pos0..pos0+M-1 against cache keys [0, pos] (causal).
 * scores = dot64(q,k) * 0.125; softmax_last_dim (max, exp(x-max), canonical sum, divide); out = canonical
 * reduction over keys of p_j * v_j. */
static void attention_rows(const czo_smollm* s, int layer, const float* Q, float* O, int M, size_t pos0) {
    int nh = s->cfg.n_heads, nkv = s->cfg.n_kv_heads, rep = nh / nkv;
    const float* KC = s->kc[layer]; const float* VC = s->vc[layer];
    size_t row = (size_t)nkv * HD;
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (pos0 + M + 64));
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int m = 0; m < M; ++m)
            for (int h = 0; h < nh; ++h) {
                int kvh = h / rep;
                size_t nk = pos0 + m + 1;
                const float* q = Q + ((size_t)m * nh + h) * HD;
                float mx = -INFINITY;
                for (size_t j = 0; j < nk; ++j) {
                    const float* k = KC + j * row + (size_t)kvh * HD;
                    float p = 0.0f;
                    for (int d = 0; d < HD; ++d) p = __builtin_fmaf(q[d], k[d], p);
                    p = p * 0.125f;
                    sc[j] = p; if (p > mx) mx = p;
                }
                for (size_t j = 0; j < nk; ++j) sc[j] = czo_expf(sc[j] - mx);
                float sum = czo_sum(sc, (int)nk);
                for (size_t j = 0; j < nk; ++j) sc[j] = sc[j] / sum;
                float tot[HD];
                for (size_t j0 = 0; j0 < nk; j0 += 64) {
                    size_t je = j0 + 64 < nk ? j0 + 64 : nk;
                    float p[HD];
                    for (int d = 0; d < HD; ++d) p[d] = 0.0f;
                    for (size_t j = j0; j < je; ++j) {
                        const float* v = VC + j * row + (size_t)kvh * HD;
                        float pj = sc[j];
                        for (int d = 0; d < HD; ++d) p[d] = __builtin_fmaf(pj, v[d], p[d]);
                    }
                    if (j0 == 0) for (int d = 0; d < HD; ++d) tot[d] = p[d];
                    else for (int d = 0; d < HD; ++d) tot[d] = tot[d] + p[d];
                }
                float* o = O + ((size_t)m * nh + h) * HD;
                for (int d = 0; d < HD; ++d) o[d] = tot[d];
            }
        free(sc);
    }
}

/* forward n tokens at positions pos0.., append K/V to the cache, logits for rows [first_row, n) */
static int llama_forward(czo_smollm* s, const uint32_t* toks, size_t n, size_t pos0, size_t first_row, float* logits) {
    const czo_llama_config* c = &s->cfg;
    int d = c->d_model, kvd = c->n_kv_heads * HD, f = c->d_ffn, M = (int)n;
    if (pos0 + n > (size_t)c->max_pos) return -1;
    for (size_t i = 0; i < n; ++i) if (toks[i] >= (uint32_t)c->vocab) return -2;
    ensure_cache(s, pos0 + n);
    float* x = (float*)malloc(sizeof(float) * n * d);
    float* xn = (float*)malloc(sizeof(float) * n * d);
    float* q = (float*)malloc(sizeof(float) * n * d);
    float* att = (float*)malloc(sizeof(float) * n * d);
    float* g = (float*)malloc(sizeof(float) * n * f);
    float* u = (float*)malloc(sizeof(float) * n * f);
    size_t emb_n; const uint16_t* emb = czo_smollm_tensor_bf16(s, "model.embed_tokens.weight", &emb_n);
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < d; ++k) x[i * d + k] = bf16_to_f32(emb[(size_t)toks[i] * d + k]);
    for (int l = 0; l < c->n_layers; ++l) {
        rms_norm_rows(x, s->ln1[l], xn, M, d, c->rms_eps);
        czo_gemm_canon(xn, d, &s->wq[l], q, d, M);
        float* kdst = s->kc[l] + pos0 * kvd; float* vdst = s->vc[l] + pos0 * kvd;
        czo_gemm_canon(xn, d, &s->wk[l], kdst, kvd, M);
        czo_gemm_canon(xn, d, &s->wv[l], vdst, kvd, M);
        rope_rows(q, M, c->n_heads, pos0, s->rope_cos, s->rope_sin);
        rope_rows(kdst, M, c->n_kv_heads, pos0, s->rope_cos, s->rope_sin);
        attention_rows(s, l, q, att, M, pos0);
        czo_gemm_canon(att, d, &s->wo[l], xn, d, M);
        for (size_t i = 0; i < n * (size_t)d; ++i) x[i] = xn[i] + x[i];          /* attn_out + residual */
        rms_norm_rows(x, s->ln2[l], xn, M, d, c->rms_eps);
        czo_gemm_canon(xn, d, &s->wg[l], g, f, M);
        czo_gemm_canon(xn, d, &s->wu[l], u, f, M);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n * (size_t)f; ++i) {
            float gv = g[i];
            float sl = gv / (1.0f + czo_expf(-gv));                               /* candle silu: x / (1 + exp(-x)) */
            g[i] = sl * u[i];
        }
        czo_gemm_canon(g, f, &s->wd[l], xn, d, M);
        for (size_t i = 0; i < n * (size_t)d; ++i) x[i] = xn[i] + x[i];          /* mlp_out + residual */
    }
    size_t nrows = n - first_row;
    rms_norm_rows(x + first_row * d, s->lnf, xn, (int)nrows, d, c->rms_eps);
    czo_gemm_canon(xn, d, &s->head, logits, c->vocab, (int)nrows);
    free(x); free(xn); free(q); free(att); free(g); free(u);
    return 0;
}

int czo_smollm_step_logits(czo_smollm* s, uint32_t token, float* logits) {          /* src/models.rs:92-103 */
    int rc = llama_forward(s, &token, 1, s->index_pos, 0, logits);
    if (rc == 0) s->index_pos += 1;
    return rc;
}
int czo_smollm_reprime_logits(czo_smollm* s, const uint32_t* hist, size_t n, float* logits) { /* src/models.rs:104-119 */
    s->index_pos = 0;                       /* fresh cache: stale rows are simply overwritten/ignored */
    if (n == 0) return -3;                  /* "reprime called with empty history" */
    int rc = llama_forward(s, hist, n, 0, n - 1, logits);
    if (rc == 0) s->index_pos += n;
    return rc;
}

/* f4 — cross_entropy_bits_over_span (src/main.rs:1725-1751): prime = tail of history (as much as fits beside the hint in
 * max_context_length() - 1 = 511 positions) ++ hint; re-prime; then for every target: floor-softmax (src/main.rs:758-766), bits +=
 * -log2(max(p, 1e-300)), step(target).  Returns 0, or the session error (-3: empty prime).  If p_out != NULL it receives the n_t
 * probabilities (what a device implementation has to reproduce bit for bit; the log2 and the running sum are host libm work). */
int czo_smollm_xe_bits_over_span(czo_smollm* s, const uint32_t* history, size_t n_hist, const uint32_t* hint, size_t n_hint,
                                 const uint32_t* targets, size_t n_t, double* bits_out, double* p_out) {
    *bits_out = 0.0;
    if (n_t == 0) return 0;                                              /* :1726 */
    const size_t max_ctx = 512 - 1;                                      /* max_context_length().saturating_sub(1), src/models.rs:91 */
    const size_t hint_budget = n_hint < max_ctx ? n_hint : max_ctx;
    const size_t remaining = max_ctx - hint_budget;
    const size_t hist_take = remaining < n_hist ? remaining : n_hist;
    uint32_t* prime = (uint32_t*)malloc(sizeof(uint32_t) * (hist_take + hint_budget + 1));
    size_t np = 0;
    for (size_t i = 0; i < hist_take; ++i) prime[np++] = history[n_hist - hist_take + i];
    for (size_t i = 0; i < hint_budget; ++i) prime[np++] = hint[i];
    const size_t V = czo_smollm_vocab(s);
    float* logits = (float*)malloc(sizeof(float) * V);
    double* pdf = (double*)malloc(sizeof(double) * V);
    int rc = czo_smollm_reprime_logits(s, prime, np, logits);
    double bits = 0.0;
    for (size_t i = 0; i < n_t && rc == 0; ++i) {
        czo_softmax_pdf_floor(logits, V, czo_ac_p_min(), pdf);
        double p = targets[i] < V ? pdf[targets[i]] : czo_ac_p_min();    /* pdf.get(sym).unwrap_or(ac_p_min()) */
        if (p_out) p_out[i] = p;
        if (!(p > 1e-300)) p = 1e-300;                                   /* p.max(1e-300) */
        bits += -log2(p);
        rc = czo_smollm_step_logits(s, targets[i], logits);
    }
    free(prime); free(logits); free(pdf);
    *bits_out = bits;
    return rc;
}
int czo_smollm_forward_rows(czo_smollm* s, const uint32_t* toks, size_t n, size_t first_row, float* logits) {
    s->index_pos = 0;
    int rc = llama_forward(s, toks, n, 0, first_row, logits);
    if (rc == 0) s->index_pos = n;
    return rc;
}

/* ======================================================================================
 * A-6 / A-7 : the reference's token loops (agent off)
 * ====================================================================================== */
size_t czo_smollm_encode_tokens(czo_smollm* s, const uint32_t* ids, size_t n, uint32_t context,
                                uint32_t reprime_interval, uint32_t* c_lo_out, uint32_t* c_hi_out,
                                uint8_t** payload, uint64_t* zero_width) {
    size_t V = czo_smollm_vocab(s);
    float* logits = (float*)malloc(sizeof(float) * V);
    uint32_t* cdf = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    czo_encoder e; czo_enc_init(&e);
    s->index_pos = 0;
    czo_smollm_step_logits(s, ids[0], logits);                                   /* src/main.rs:1916 */
    size_t effective_context = context < 511 ? context : 511;                    /* :1935 */
    for (size_t i = 0; i + 1 < n; ++i) {                                         /* :1979 ids.iter().skip(1).enumerate() */
        uint32_t sym = ids[i + 1];
        if (s->index_pos >= effective_context && (i % reprime_interval) == 0 && i > 0) {   /* :2280 */
            size_t end = 1 + i;
            size_t start = end > effective_context ? end - effective_context : 0;
            czo_smollm_reprime_logits(s, ids + start, end - start, logits);      /* :2281-2284 */
        }
        czo_logits_to_cdf_smollm(logits, V, cdf);                                /* :2294-2296 */
        if (c_lo_out) c_lo_out[i] = cdf[sym];
        if (c_hi_out) c_hi_out[i] = cdf[sym + 1];
        czo_enc_encode_counts(&e, cdf[sym], cdf[sym + 1], CZO_AC_CDF_TOTAL);     /* :2297-2300 */
        czo_smollm_step_logits(s, sym, logits);                                  /* :2345 */
    }
    const uint8_t* out; size_t len = czo_enc_finish(&e, &out);                   /* :2358 */
    if (payload) { *payload = (uint8_t*)malloc(len ? len : 1); memcpy(*payload, out, len); }
    if (zero_width) *zero_width = e.zero_width;
    czo_enc_free(&e); free(logits); free(cdf);
    return len;
}

int czo_smollm_decode_tokens(czo_smollm* s, const uint8_t* payload, size_t len, size_t token_count,
                             uint32_t context_window, uint32_t reprime_interval, uint32_t bos,
                             uint32_t* out_ids) {
    size_t V = czo_smollm_vocab(s);
    float* logits = (float*)malloc(sizeof(float) * V);
    uint32_t* cdf = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    czo_decoder d; czo_dec_init(&d, payload, len);                               /* src/main.rs:2485 */
    s->index_pos = 0;
    czo_smollm_step_logits(s, bos, logits);
    size_t n_out = 0; out_ids[n_out++] = bos;
    size_t ec = context_window > 0 ? context_window - 1 : 0;                     /* :2506 saturating_sub(1).min(511) */
    if (ec > 511) ec = 511;
    for (size_t i = 0; i < token_count; ++i) {                                   /* :2528 */
        if (s->index_pos >= ec && (i % reprime_interval) == 0 && i > 0) {        /* :2532 */
            size_t end = n_out, start = end > ec ? end - ec : 0;
            czo_smollm_reprime_logits(s, out_ids + start, end - start, logits);  /* :2533-2536 */
        }
        czo_logits_to_cdf_smollm(logits, V, cdf);                                /* :2622-2624 */
        uint32_t sym = czo_dec_decode_symbol_counts(&d, cdf, V, CZO_AC_CDF_TOTAL); /* :2625 */
        out_ids[n_out++] = sym;
        czo_smollm_step_logits(s, sym, logits);                                  /* :2627 */
    }
    free(logits); free(cdf);
    return 0;
}


/* ======================================================================================
 * per-op entry points for kernel-level parity tests (same code paths as the model above)
 * ====================================================================================== */
void czo_gemm_bf16(const float* A, const uint16_t* W, const float* R, float* C, int M, int N, int K) {
    packed_w pw; czo_pack_weights(&pw, W, N, K);
    czo_gemm_canon(A, K, &pw, C, N, M);
    if (R) for (size_t i = 0; i < (size_t)M * N; ++i) C[i] = C[i] + R[i];
    free(pw.p);
}
void czo_rmsnorm_bf16w(const float* X, const uint16_t* w, float* Y, int M, int d, float eps) {
    float* wf = widen(w, d);
    rms_norm_rows(X, wf, Y, M, d, eps);
    free(wf);
}
/* rows at positions pos0.. attend to K/V rows [0, pos0 + m]; Q [M][nh*64], K/V [nk][nkv*64], O [M][nh*64] */
void czo_attention_ref(const float* Q, const float* K, const float* V, float* O, int M, int pos0, int nh, int nkv) {
    czo_smollm s; memset(&s, 0, sizeof s);
    s.cfg.n_heads = nh; s.cfg.n_kv_heads = nkv;
    float* kc = (float*)K; float* vc = (float*)V;
    s.kc = &kc; s.vc = &vc;
    attention_rows(&s, 0, Q, O, M, (size_t)pos0);
}
void czo_swiglu(const float* GU, float* H, int M, int f) {
    for (int m = 0; m < M; ++m) for (int j = 0; j < f; ++j) {
        float g = GU[(size_t)m * 2 * f + j], u = GU[(size_t)m * 2 * f + f + j];
        H[(size_t)m * f + j] = (g / (1.0f + czo_expf(-g))) * u;
    }
}

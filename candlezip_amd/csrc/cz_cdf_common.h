// cz_cdf_common.h — device functions shared by the serial (kernels_cdf.hip) and exact-parallel (kernels_cdf_par.hip)
// CDF / arithmetic-decoder kernels.  Restates reference src/main.rs:758-824 (pdf, cdf) and 419-449, 500-548 (decoder).
#pragma once
#include "cz_kernels.h"

#define CDF_TOTAL 1073741824.0   // 2^30 (AC_CDF_TOTAL, src/main.rs:803)
#define CDF_TOTAL_U 1073741824u

__device__ __forceinline__ uint32_t quantize(double acc, uint32_t prev) {   // src/main.rs:812-820
    double f = floor(acc * CDF_TOTAL);
    long long v;
    if (f != f) v = 0;                                   // Rust `as i64`: NaN -> 0, saturating
    else if (f >= 9.2233720368547758e18) v = 0x7fffffffffffffffLL;
    else if (f <= -9.2233720368547758e18) v = (long long)0x8000000000000000ULL;
    else v = (long long)f;
    if (v < 0) v = 0;
    if (v > (long long)CDF_TOTAL_U) v = (long long)CDF_TOTAL_U;
    if (v < (long long)prev) v = (long long)prev;
    return (uint32_t)v;
}

// a / b correctly rounded (== the IEEE division the reference does, bit for bit) for a divisor b that a whole row shares: y = 1.0 / b is
// taken ONCE (a true division), every quotient is then one multiply and four fused multiply-adds instead of the ~11-instruction division
// sequence with its quarter-rate v_rcp_f64.  Why it is exact (round to nearest, no under-/overflow: here 0 <= a <= 1 or a few units, 1 <= b
// < 2^17, quotients above 2^-170):  y = RN(1/b), so q0 = RN(a y) is within 1.5 ulp of a/b;  r0 = RN(a - b q0) and q1 = RN(q0 + r0 y)
// leave q1 within 1/2 ulp + 2^-51 ulp of a/b, i.e. FAITHFUL;  for a faithful q1 the remainder r1 = a - b q1 is a floating-point
// number (the fma computes it exactly) and Markstein's theorem (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic,
// "division with an FMA") gives RN(q1 + r1 y) = RN(a / b) whenever y is the correctly rounded reciprocal.  a = 0 gives +0 like 0 / b.
// Checked against the division itself: tests/test_div_by_reciprocal.py (exact rational model, quotients next to rounding midpoints,
// divisors with all-ones significands) and test_gpu_kernels.py::test_division_by_shared_reciprocal_is_exact (the device function).
__device__ __forceinline__ double div_by(double a, double b, double y) {
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// per-row constants of the pdf; pdf_at() then yields pdf[i] exactly as the reference computes it
struct pdf_ctx {
    const float* l; int V, n, mode; float mx; double sum, norm, sum2, scale, lit; bool uniform;
};
#define P_MIN 1.862645149230957e-09      // ac_p_min() = 2 * 2^-30  (src/main.rs:235-238)

__device__ __forceinline__ double pdf_floor_at(const pdf_ctx& c, int i) {   // softmax_pdf_floor :763-766 (before /norm)
    double p = (double)cz_expf(c.l[i] - c.mx) / c.sum;
    return p > P_MIN ? p : P_MIN;
}
__device__ __forceinline__ double pdf_at(const pdf_ctx& c, int i) {
    if (c.mode == 0) {
        if (c.uniform) return 1.0 / (double)c.V;
        return (double)cz_expf(c.l[i] - c.mx) / c.sum;
    }
    double x = i < c.V ? (pdf_floor_at(c, i) / c.norm) * c.scale : c.lit;
    return c.sum2 > 0.0 ? x / c.sum2 : x;
}
__device__ inline void pdf_prepare(pdf_ctx& c, const float* l, int V, int mode) {
    c.l = l; c.V = V; c.mode = mode; c.n = mode == 0 ? V : V + 256; c.uniform = false;
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) { float v = l[i]; if (v > mx) mx = v; }
    c.mx = mx;
    double sum = 0.0;
    for (int i = 0; i < V; ++i) sum += (double)cz_expf(l[i] - mx);
    c.sum = sum;
    if (mode == 0) { c.uniform = sum <= 0.0; return; }
    double norm = 0.0;
    for (int i = 0; i < V; ++i) norm += pdf_floor_at(c, i);
    c.norm = norm;
    const double p_escape_total = 256.0 * P_MIN;             // :771
    double scale = 1.0 - p_escape_total; if (!(scale > 0.0)) scale = 0.0;
    c.scale = scale;
    c.lit = p_escape_total > 0.0 ? p_escape_total / 256.0 : 0.0;
    double s2 = 0.0;
    for (int i = 0; i < V; ++i) s2 += (pdf_floor_at(c, i) / norm) * scale;
    for (int i = 0; i < 256; ++i) s2 += c.lit;
    c.sum2 = s2;
}

// ---------------- arithmetic decoder on the device ----------------
#define AC_FULL 0xFFFFFFFFull
#define AC_HALF 0x80000000ull
#define AC_QUARTER 0x40000000ull

__device__ __forceinline__ unsigned dec_get_bit(cz_dec_state& d) {          // :438-449 ; EOF reads as 1 (:432, :545)
    if (d.byte_pos >= d.in_len) { if (!d.final_input) d.underrun = 1u; return 1u; }
    unsigned byte = d.in[d.byte_pos];
    unsigned bit = (byte >> (7 - d.bit_pos)) & 1u;
    d.bit_pos += 1;
    if (d.bit_pos >= 8) { d.bit_pos = 0; d.byte_pos += 1; }
    return bit;
}
// low 64 bits of floor((nh:nl) / d), d != 0
__device__ inline unsigned long long div128_lo64(unsigned long long nh, unsigned long long nl, unsigned long long d) {
    if (nh == 0) return nl / d;
    unsigned long long q = 0, r = 0;
    for (int b = 127; b >= 0; --b) {
        const unsigned long long bit = b >= 64 ? (nh >> (b - 64)) & 1ull : (nl >> b) & 1ull;
        const unsigned long long carry = r >> 63;
        r = (r << 1) | bit;
        if (carry || r >= d) { r -= d; if (b < 64) q |= 1ull << b; }
    }
    return q;
}
// the symbol-independent part of decode_symbol_counts: value (:503-505) -- and, given the symbol's interval, the update (:516-546)
__device__ __forceinline__ uint32_t dec_value(const cz_dec_state& d, bool& ok) {
    const unsigned long long range = d.high - d.low + 1;                     // u64, wrapping
    if (range == 0) { ok = false; return 0; }                               // the reference would panic (divide by zero)
    const unsigned long long x = d.code - d.low + 1;                         // u64, wrapping
    // (x * 2^30 - 1) in 128 bits
    unsigned long long nh = x >> 34, nl = x << 30;
    if (nl == 0) { nh -= 1; }                                                // borrow (wraps like u128 when x == 0)
    nl -= 1;
    ok = true;
    return (uint32_t)div128_lo64(nh, nl, range);
}
__device__ __forceinline__ void dec_update(cz_dec_state& d, uint32_t c_lo, uint32_t c_hi) {
    const unsigned long long range = d.high - d.low + 1;
    // (range * c) / 2^30 in 128 bits, truncated to 64 after adding low (the reference truncates with `as u64`)
    const unsigned long long lo_lo = range * (unsigned long long)c_lo, lo_hi = __umul64hi(range, (unsigned long long)c_lo);
    const unsigned long long hi_lo = range * (unsigned long long)c_hi, hi_hi = __umul64hi(range, (unsigned long long)c_hi);
    const unsigned long long ql = (lo_lo >> 30) | (lo_hi << 34);
    const unsigned long long qh = (hi_lo >> 30) | (hi_hi << 34);
    const unsigned long long low0 = d.low;
    d.low = low0 + ql;
    d.high = low0 + qh - 1;
    for (;;) {
        if (d.high < AC_HALF) {
        } else if (d.low >= AC_HALF) {
            d.low -= AC_HALF; d.high -= AC_HALF; d.code -= AC_HALF;
        } else if (d.low >= AC_QUARTER && d.high < AC_QUARTER * 3) {
            d.low -= AC_QUARTER; d.high -= AC_QUARTER; d.code -= AC_QUARTER;
        } else break;
        d.low = (d.low << 1) & AC_FULL;
        d.high = ((d.high << 1) & AC_FULL) | 1;
        d.code = ((d.code << 1) & AC_FULL) | dec_get_bit(d);
    }
}


// one row, serial walk (the reference's own order): coder interval of symbol s
__device__ __attribute__((noinline)) void serial_pairs_row(const float* l, int V, int mode, uint32_t s, uint32_t* lo_out, uint32_t* hi_out) {
    pdf_ctx c;
    pdf_prepare(c, l, V, mode);
    double acc = 0.0;
    uint32_t prev = 0, lo = 0, hi = 0;
    for (uint32_t i = 0; i <= s && i < (uint32_t)c.n; ++i) {
        acc += pdf_at(c, (int)i);
        uint32_t v = quantize(acc, prev);
        if (i + 1 == (uint32_t)c.n) v = CDF_TOTAL_U;       // cdf[n] = total forced (:822)
        if (i + 1 == s) lo = v;
        if (i == s) hi = v;
        prev = v;
    }
    *lo_out = lo; *hi_out = hi;
}
// one row, serial walk: decode one symbol and update the decoder state; returns false if the reference would panic
__device__ __attribute__((noinline)) bool serial_decode_row(const float* l, int V, int mode, cz_dec_state& d, uint32_t* s_out) {
    pdf_ctx c;
    pdf_prepare(c, l, V, mode);
    bool ok;
    const uint32_t value = dec_value(d, ok);
    if (!ok) return false;
    // symbol s with cdf[s] <= value < cdf[s+1] (the reference's binary search :508-514; unique, so a scan finds the same s;
    // value >= total falls through to the last symbol exactly like the search does)
    double acc = 0.0;
    uint32_t prev = 0, lo = 0, hi = 0, s = (uint32_t)c.n - 1;
    for (uint32_t k = 0; k < (uint32_t)c.n; ++k) {
        acc += pdf_at(c, (int)k);
        uint32_t v = quantize(acc, prev);
        if (k + 1 == (uint32_t)c.n) v = CDF_TOTAL_U;
        if (v > value || k + 1 == (uint32_t)c.n) { s = k; lo = prev; hi = v; break; }
        prev = v;
    }
    dec_update(d, lo, hi);
    *s_out = s;
    return true;
}

// cz_numerics.h — scalar arithmetic shared by host and device code of libcandlezip_hip.
//
// The reference calls the platform libm (f32::exp at src/main.rs:790; candle's exp/silu/sigmoid/tanh/ln),
// which is not reproducible across hosts.  The product owns ONE definition of each
// transcendental, built from IEEE +,*,fma and integer operations only, so that the HIP kernels and any CPU
// decoder of the same format agree bit for bit.  Must be compiled with -ffp-contract=off: every fused
// multiply-add is written explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CZ_HD __host__ __device__ __forceinline__

CZ_HD float cz_bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
CZ_HD uint32_t cz_f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
CZ_HD float cz_bf16_to_f32(uint16_t b) { return cz_bits_f32((uint32_t)b << 16); }
CZ_HD uint16_t cz_f32_to_bf16_rne(float f) {
    uint32_t u = cz_f32_bits(f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
CZ_HD float cz_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exp: Cody-Waite reduction by ln2 (hi/lo), degree-5 polynomial, two-step 2^n scaling (one rounding for
// subnormal results).  exp(-inf) = 0, exp(x > 88.73) = +inf, NaN propagates.
// The definition (host, oracle, and the yardstick of the device form below).
CZ_HD float cz_expf_ref(float x) {
    if (x != x) return x;
    if (x < -104.0f) x = -104.0f;
    if (x > 89.0f) x = 89.0f;
    const float MAGIC = 12582912.0f;
    float t = cz_fma(x, 1.44269504088896341f, MAGIC);
    float n = t - MAGIC;
    int ni = (int)n;
    float r = cz_fma(n, -0.693145751953125f, x);
    r = cz_fma(n, -1.42860682030941723e-06f, r);
    float p = 1.9875691500e-4f;
    p = cz_fma(p, r, 1.3981999507e-3f);
    p = cz_fma(p, r, 8.3334519073e-3f);
    p = cz_fma(p, r, 4.1665795894e-2f);
    p = cz_fma(p, r, 1.6666665459e-1f);
    p = cz_fma(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = cz_fma(p, r2, r);
    y = y + 1.0f;
    int k1 = ni / 2, k2 = ni - k1;
    float s1 = cz_bits_f32((uint32_t)(k1 + 127) << 23);
    float s2 = cz_bits_f32((uint32_t)(k2 + 127) << 23);
    return (y * s1) * s2;
}
// Device form, the SAME function (round 4): the clamps as v_max / v_min (a NaN is put back by the last select), the two-step 2^n scaling as ONE
// v_ldexp_f32 — y * 2^k1 is exact (k1 in [-75, 64], y in [0.5, 2.8)) and the second product rounds once, which is what ldexp does — 20 vector
// instructions instead of 31.  A vector instruction costs the matrix pipe of its SIMD ~3 cycles (35 beside another wave's MFMA stream), and the
// GEMM's SwiGLU epilogue, the prefill attention and the CDF kernels are made of these.  tests/test_gpu_round4.py compares the two forms on ALL
// 2^32 float bit patterns on the device (cz_debug_expf_exhaustive): zero differences.
CZ_HD float cz_expf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float xc = __builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.0f);
    const float MAGIC = 12582912.0f;
    float t = cz_fma(xc, 1.44269504088896341f, MAGIC);
    float n = t - MAGIC;
    int ni = (int)n;
    float r = cz_fma(n, -0.693145751953125f, xc);
    r = cz_fma(n, -1.42860682030941723e-06f, r);
    float p = 1.9875691500e-4f;
    p = cz_fma(p, r, 1.3981999507e-3f);
    p = cz_fma(p, r, 8.3334519073e-3f);
    p = cz_fma(p, r, 4.1665795894e-2f);
    p = cz_fma(p, r, 1.6666665459e-1f);
    p = cz_fma(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = cz_fma(p, r2, r);
    y = y + 1.0f;
    const float res = __builtin_amdgcn_ldexpf(y, ni);
    return x != x ? x : res;
#else
    return cz_expf_ref(x);
#endif
}

CZ_HD float cz_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return cz_bits_f32(0x7fc00000u);
    if (x == 0.0f) return cz_bits_f32(0xff800000u);
    if (x == cz_bits_f32(0x7f800000u)) return x;
    uint32_t u = cz_f32_bits(x);
    int e = 0;
    if ((u >> 23) == 0) { x = x * 8388608.0f; u = cz_f32_bits(x); e = -23; }
    e += (int)(u >> 23) - 126;
    float m = cz_bits_f32((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = cz_fma(p, m, -1.1514610310e-1f);
    p = cz_fma(p, m, 1.1676998740e-1f);
    p = cz_fma(p, m, -1.2420140846e-1f);
    p = cz_fma(p, m, 1.4249322787e-1f);
    p = cz_fma(p, m, -1.6668057665e-1f);
    p = cz_fma(p, m, 2.0000714765e-1f);
    p = cz_fma(p, m, -2.4999993993e-1f);
    p = cz_fma(p, m, 3.3333331174e-1f);
    float y = (m * z) * p;
    float fe = (float)e;
    y = cz_fma(fe, -2.12194440e-4f, y);
    y = cz_fma(-0.5f, z, y);
    float res = m + y;
    res = cz_fma(fe, 0.693359375f, res);
    return res;
}

CZ_HD float cz_tanhf(float x) {
    if (x != x) return x;
    float ax = x < 0.0f ? -x : x;
    if (ax >= 9.0f) return x < 0.0f ? -1.0f : 1.0f;
    if (ax >= 0.625f) {
        float e = cz_expf(ax + ax);
        float r = 1.0f - 2.0f / (e + 1.0f);
        return x < 0.0f ? -r : r;
    }
    float z = x * x;
    float p = -5.70498872745e-3f;
    p = cz_fma(p, z, 2.06390887954e-2f);
    p = cz_fma(p, z, -5.37397155531e-2f);
    p = cz_fma(p, z, 1.33314422036e-1f);
    p = cz_fma(p, z, -3.33332819422e-1f);
    p = p * z;
    return cz_fma(p, x, x);
}

CZ_HD float cz_sigmoidf(float x) { return 1.0f / (cz_expf(-x) + 1.0f); }   // candle sigmoid: (exp(-x) + 1).recip()
CZ_HD float cz_siluf(float x) { return x / (1.0f + cz_expf(-x)); }          // candle silu: x / (1 + exp(-x))

// Synthetic-weight generator (DESIGN.md "Synthetic mode"): bf16 bits of element idx of tensor tid.
CZ_HD uint64_t cz_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
CZ_HD uint16_t cz_synth_bf16(uint64_t seed, uint32_t tid, uint64_t idx, float stdv, float mean) {
    uint64_t h = cz_splitmix64(seed ^ cz_splitmix64(((uint64_t)tid << 32) ^ idx));
    int s = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48) - 131070;
    float scale = stdv * 2.6428893e-05f;
    float v = (float)s * scale;
    v = v + mean;
    return cz_f32_to_bf16_rne(v);
}
static inline uint32_t cz_fnv1a32(const char* s) {
    uint32_t h = 2166136261u;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 16777619u; }
    return h;
}

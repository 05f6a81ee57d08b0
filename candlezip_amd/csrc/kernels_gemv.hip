// kernels_gemv.hip — the projections of a decode step for a HANDFUL of streams (1..4 rows: the CLI's one stream per file), on the VALU.
//
// dec_gemm_kernel (kernels_decode.hip) serves these groups with 16-row MFMA tiles: fifteen of sixteen rows are padding, and a block is a chain
// of five stages of one memory round trip or one block barrier each — first operand, norm phase, MFMA chains of nine to twelve waves, ordered
// reduction through LDS behind two barriers, epilogue: 5-9 us per launch (profiles/r03_dec_phases_b1.json), 155 launches per token.  For one
// row the canonical arithmetic needs no matrix core: an output column is nk chunk chains  p = fmaf(x_k, w_k, p), k ascending from +0 — the very
// chain an MFMA runs — and the ordered sum of the chunk partials.  Here
//   * block = ONE wave, no block barrier anywhere; lane = (unit, chunk group): the LPC = ceil(nk / CPL) lanes of a unit own CPL consecutive
//     64-chunks each of the unit's one or two weight rows (two: the rotation pair (j, j + 32) of a head / the gate and up rows of a hidden column,
//     so that the epilogue finds both values in one lane);
//   * every weight of the lane (CPL x 128 B per row, natural [N][K] layout: a unit's lanes read one contiguous row) and the epilogue's operands are
//     requested at the top; the activation row — normalised on the way where the projection follows an RMSNorm (the wave redoes the norm of
//     its <= 4 rows: nine 64-long chains side by side on nine lanes) — waits in LDS, chunk c at c * 68 floats (a b128 read of nine chunks at
//     once touches nine different bank quads);
//   * the chunk partials meet in LDS (a wave's own lanes: no barrier) and the unit's first lane adds them left to right.
// Same chains, same order: the same bits as dec_gemm_kernel / gemm_bf16w_kernel / the oracle.  Bound: launch + one memory round trip (HBM-cold
// weights, 16-24 KB per wave).
#include "cz_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { GV_PLAIN = 0, GV_ROPE = 1, GV_SWIGLU = 2, GV_RELUSQ = 3 };      // GV_RELUSQ: C = relu(A.W^T)^2 (the RWKV7 channel-mix key projection, rwkv7.rs:426)
#define GV_CS 68                            // floats per 64-chunk of an activation row in LDS
struct gemv_args {
    const float* A; int lda;                // natural layout [M][K] (NORM: the residual stream, RMSNorm(normw, eps) applied on the way)
    const uint16_t* W;                      // natural bf16 [N][K]; GV_SWIGLU: [gate rows (f) | up rows (f)]
    int M, N, K;
    float* C; int ldc; const float* R;      // GV_PLAIN: C = A.W^T (+ R, may alias C); GV_SWIGLU: C [M][f] = silu(gate) * up; GV_ROPE: q -> C
    const unsigned char* row_mask;          // GV_PLAIN: rows with mask == 0 are not stored (RWKV literal symbols do not step the model)
    const uint16_t* normw; float eps;
    int n_heads, n_kv; const cz_rope_desc* descs; size_t kv_off; const int* pos_dev; const float* cs; const float* sn;   // GV_ROPE
    int nunits;                             // PLAIN: N columns; ROPE: N / 2 pairs; SWIGLU: f pairs
    // ASUM (GV_PLAIN, K = n_heads x 64): A is not read — element (head h, d) of row r is the ordered sum over the key chunks of the split decode
    // attention's P.V partials, part[((r n_heads + h) part_maxc + c) 64 + d], c = 0 .. ceil((*pos_dev + 1) / 64) - 1: what da_split_sum_kernel
    // would have written, without its launch (kernels_decode.hip, "Split decode attention")
    const float* part; int part_maxc;
};

template <int EPI, bool NORM, int MB, int CPL, int ASUM, int XRP>
__device__ __forceinline__ void dec_gemv_body(const gemv_args& a, const int bx, const int by) {
    static_assert(ASUM == 0 || (EPI == GV_PLAIN && !NORM && MB == 1 && CPL == 1), "ASUM: the o-projection of one row per block");
    constexpr int NC = (EPI == GV_PLAIN || EPI == GV_RELUSQ) ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int K = a.K, nk = K >> 6, LPC = (nk + CPL - 1) / CPL, upw = 64 / LPC;
    const int XS = nk * GV_CS;                                 // floats per activation row
    float* xs = smem;                                          // [MB][XS]
    float* part = smem + MB * XS;                              // [64 lanes][CPL][NC][MB]
    float* ssq = part + 64 * CPL * NC * MB;                    // [MB][nk] chunk sums of squares (NORM)
    const int ul = lane / LPC, cg = lane - ul * LPC;
    const bool lane_on = ul < upw;
    const int unit_raw = bx * upw + ul;
    const bool unit_on = lane_on && unit_raw < a.nunits;
    const int unit = unit_raw < a.nunits ? unit_raw : a.nunits - 1;     // (clamped: idle lanes load what a neighbour loads, never a conditional load)
    const int m0 = by * MB;
    int n[NC];
    if (NC == 1) n[0] = unit;
    else if (EPI == GV_ROPE) { n[0] = (unit >> 5) * 64 + (unit & 31); n[NC - 1] = n[0] + 32; }
    else { n[0] = unit; n[NC - 1] = a.N / 2 + unit; }
    // ---- everything that depends on no result, requested now: the lane's weights, the epilogue's operands, the activation rows
    u32x4 w[CPL][NC][8];
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
        const int c = min(cg * CPL + ci, nk - 1);
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const u32x4* src = reinterpret_cast<const u32x4*>(a.W + (size_t)n[e] * K + c * 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) w[ci][e][j] = src[j];
        }
    }
    float res[MB];
    int pos = 0; float rc = 0.0f, rs = 0.0f; cz_rope_desc rd[MB];
#pragma unroll
    for (int r = 0; r < MB; ++r) res[r] = 0.0f;
    if (EPI == GV_PLAIN && a.R) {
#pragma unroll
        for (int r = 0; r < MB; ++r) res[r] = a.R[(size_t)min(m0 + r, a.M - 1) * a.ldc + n[0]];
    }
    if (EPI == GV_ROPE) {
#pragma unroll
        for (int r = 0; r < MB; ++r) rd[r] = a.descs[min(m0 + r, a.M - 1)];
        pos = a.pos_dev ? *a.pos_dev : rd[0].pos0;
        rc = a.cs[(size_t)pos * 32 + (unit & 31)]; rs = a.sn[(size_t)pos * 32 + (unit & 31)];
    }
    constexpr int XR = XRP ? XRP : (CPL == 1 ? 12 : 24);       // activation elements per lane and row (launcher: nk <= 12 at CPL 1, <= 24 / 48 at CPL 3)
    float xv[MB][XR];
    float nw[NORM ? XR : 1];
    if (ASUM) {
        // every partial of the row in flight at once (a loop on the runtime chunk count would be a chain of L2 round trips), then head by head the chunks left to right
        const int nch = ((a.pos_dev ? *a.pos_dev : 0) + 1 + 63) >> 6;
        float pv[XR][ASUM ? ASUM : 1];
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            const float* pp = a.part + ((size_t)m0 * nk + min(j, nk - 1)) * a.part_maxc * 64 + lane;      // (nk = n_heads)
#pragma unroll
            for (int c = 0; c < (ASUM ? ASUM : 1); ++c) pv[j][c] = pp[(size_t)min(c, nch - 1) * 64];
        }
#pragma unroll
        for (int j = 0; j < XR; ++j) {
            float t = pv[j][0];
#pragma unroll
            for (int c = 1; c < (ASUM ? ASUM : 1); ++c) if (c < nch) t = t + pv[j][c];
            xv[0][j] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < XR; ++j) {                         // (clamped addresses, never a conditional load; elements past K are never used)
            const int k = min(lane + 64 * j, K - 1);
#pragma unroll
            for (int r = 0; r < MB; ++r) xv[r][j] = a.A[(size_t)min(m0 + r, a.M - 1) * a.lda + k];
            if (NORM) nw[NORM ? j : 0] = cz_bf16_to_f32(a.normw[k]);
        }
    }
    // ---- the rows into LDS (chunk j of a row = elements 64 j + lane: one b32 write per chunk and lane)
#pragma unroll
    for (int r = 0; r < MB; ++r)
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (j < nk) xs[r * XS + j * GV_CS + lane] = xv[r][j];
    if (NORM) {
        // candle rms_norm: m = sqrt(sum(x*x)/d + eps), the sum in canonical 64-chunks (squares rounded, added k ascending, chunk sums left to right); y = x / m * w.
        // lane r * nk + c walks chunk c of row r (MB * nk <= 64 lanes: launcher)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        {
            const int r = lane / nk, c = lane - r * nk;
            if (r < MB) {
                const float* src = xs + r * XS + c * GV_CS;
                float p = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);
                    const float s0 = v[0] * v[0], s1 = v[1] * v[1], s2 = v[2] * v[2], s3 = v[3] * v[3];
                    p = p + s0; p = p + s1; p = p + s2; p = p + s3;
                }
                ssq[r * nk + c] = p;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int r = 0; r < MB; ++r) {
            float t = ssq[r * nk];
            for (int c = 1; c < nk; ++c) t = t + ssq[r * nk + c];
            const float mv = sqrtf(t / (float)K + a.eps);
#pragma unroll
            for (int j = 0; j < XR; ++j)
                if (j < nk) xs[r * XS + j * GV_CS + lane] = xv[r][j] / mv * nw[NORM ? j : 0];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- the chunk chains: p = fmaf(x_k, w_k, p), k ascending from +0 (== one MFMA chain of the matrix-core kernels)
    float p[CPL][NC][MB];
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
        const int c = min(cg * CPL + ci, nk - 1);
#pragma unroll
        for (int e = 0; e < NC; ++e)
#pragma unroll
            for (int r = 0; r < MB; ++r) p[ci][e][r] = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x4 x4[MB];
#pragma unroll
            for (int r = 0; r < MB; ++r) x4[r] = *reinterpret_cast<const f32x4*>(xs + r * XS + c * GV_CS + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 4 * q + t;                                    // element k of the chunk: half (k & 1) of word (k >> 1) & 3 of piece k >> 3
#pragma unroll
                for (int e = 0; e < NC; ++e) {
                    const unsigned ww = w[ci][e][k >> 3][(k >> 1) & 3];
                    const float wv = (k & 1) ? cz_bits_f32(ww & 0xffff0000u) : cz_bits_f32(ww << 16);
#pragma unroll
                    for (int r = 0; r < MB; ++r) p[ci][e][r] = cz_fma(x4[r][t], wv, p[ci][e][r]);
                }
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
        for (int e = 0; e < NC; ++e)
#pragma unroll
            for (int r = 0; r < MB; ++r) part[((lane * CPL + ci) * NC + e) * MB + r] = p[ci][e][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (!unit_on || cg != 0) return;
    // ---- the unit's first lane: chunk partials left to right, epilogue
    float tot[NC][MB];
    for (int c = 0; c < nk; ++c) {
        const int src_lane = lane + c / CPL, ci = c - (c / CPL) * CPL;
#pragma unroll
        for (int e = 0; e < NC; ++e)
#pragma unroll
            for (int r = 0; r < MB; ++r) {
                const float v = part[((src_lane * CPL + ci) * NC + e) * MB + r];
                tot[e][r] = c == 0 ? v : tot[e][r] + v;
            }
    }
#pragma unroll
    for (int r = 0; r < MB; ++r) {
        const int row = m0 + r;
        if (row >= a.M) break;
        const float t0 = tot[0][r], t1 = tot[NC - 1][r];
        if (EPI == GV_PLAIN) {
            if (a.row_mask && !a.row_mask[row]) continue;
            a.C[(size_t)row * a.ldc + n[0]] = a.R ? t0 + res[r] : t0;
        } else if (EPI == GV_RELUSQ) {
            const float z = t0 > 0.0f ? t0 : 0.0f;
            a.C[(size_t)row * a.ldc + n[0]] = z * z;
        } else if (EPI == GV_SWIGLU) {
            a.C[(size_t)row * a.ldc + unit] = cz_siluf(t0) * t1;                        // silu(gate) * up (llama.rs MLP)
        } else {
            const cz_rope_desc& d = rd[r];
            const int head = unit >> 5, i = unit & 31;                                   // rotation index i in [0, 32)
            if (head < a.n_heads + a.n_kv) {
                const float aa = t0 * rc, bb = t1 * rs, c2 = t0 * rs, dd = t1 * rc;      // candle rope: products rounded separately
                const float o1 = aa - bb, o2 = c2 + dd;
                if (head < a.n_heads) {
                    a.C[(size_t)row * a.ldc + head * 64 + i] = o1; a.C[(size_t)row * a.ldc + head * 64 + i + 32] = o2;
                } else if (d.kt_stride) {
                    float* kd = d.k_dst + a.kv_off + (size_t)((head - a.n_heads) * 64) * d.kt_stride + pos;
                    kd[(size_t)i * d.kt_stride] = o1; kd[(size_t)(i + 32) * d.kt_stride] = o2;
                } else {
                    float* kd = d.k_dst + a.kv_off + (size_t)pos * d.dst_stride + (head - a.n_heads) * 64;
                    kd[i] = o1; kd[i + 32] = o2;
                }
            } else {
                float* vd = d.v_dst + a.kv_off + (size_t)pos * d.dst_stride + (head - a.n_heads - a.n_kv) * 64;
                vd[i] = t0; vd[i + 32] = t1;
            }
        }
    }
}

template <int EPI, bool NORM, int MB, int CPL, int ASUM = 0, int XRP = 0>
__global__ __launch_bounds__(64) void dec_gemv_kernel(gemv_args a) { dec_gemv_body<EPI, NORM, MB, CPL, ASUM, XRP>(a, blockIdx.x, blockIdx.y); }
// Several independent plain projections of one step in ONE launch (blockIdx.z = problem; K <= 768 each): the RWKV7 time-mix's seven
// projections of differently mixed inputs, then its four LoRA up-projections.  Blocks past a problem's own unit count leave at once.
#define GV_GROUP_MAX 8
struct gemv_group { gemv_args p[GV_GROUP_MAX]; };
template <int MB>
__global__ __launch_bounds__(64) void dec_gemv_group_kernel(gemv_group g) {
    const gemv_args& a = g.p[blockIdx.z];
    const int nk = a.K >> 6, upw = 64 / nk;
    if ((int)blockIdx.x * upw >= a.nunits) return;
    dec_gemv_body<GV_PLAIN, false, MB, 1, 0, 0>(a, blockIdx.x, blockIdx.y);
}

int g_gemv_fold_attn = 1;       // tuning knob (cz_debug_set "gemv_fold_attn"): 1 = the o-projection adds up the split attention's chunk partials itself (no da_split_sum_kernel launch)
// tuning knob (cz_debug_set "gemv_max_rows"): groups of at most this many streams take the VALU projections (0 = never).  Measured at 767 keys, ms per step, MFMA
// projections (dec_gemm_kernel) against these (profiles/r04_single_stream/): 1 stream 1.376 / 0.921, 2: 1.42 / 1.04, 4: 1.47 / 1.29, 8: 1.52 / 1.40,
// 12: 1.67 / 1.52, 16: 1.81 / 1.69, 32: 1.79 / 1.95, 64: 2.08 / 2.63 — rows beyond four are further blocks (blockIdx.y) that re-read the weights from L2
int g_gemv_max_rows = 16;
bool cz_dec_gemv_ok(int M, int K, bool norm) {
    if (g_gemv_max_rows <= 0 || M > g_gemv_max_rows || K % 64 || K > 3072) return false;      // (more than four rows: row groups of four, blockIdx.y)
    return !norm || 4 * (K >> 6) <= 64;
}
template <int EPI, bool NORM, int MB, int CPL, int ASUM = 0, int XRP = 0>
static hipError_t gemv_launch_one(hipStream_t st, gemv_args a) {
    constexpr int NC = (EPI == GV_PLAIN || EPI == GV_RELUSQ) ? 1 : 2;
    const int nk = a.K >> 6, LPC = (nk + CPL - 1) / CPL, upw = 64 / LPC;
    const size_t smem = (size_t)(MB * nk * GV_CS + 64 * CPL * NC * MB + MB * nk) * sizeof(float);
    hipLaunchKernelGGL((dec_gemv_kernel<EPI, NORM, MB, CPL, ASUM, XRP>), dim3((a.nunits + upw - 1) / upw, (a.M + MB - 1) / MB), dim3(64), smem, st, a);
    return hipGetLastError();
}
template <int EPI, bool NORM>
static hipError_t gemv_launch(hipStream_t st, const gemv_args& a) {
    // one chunk per lane up to twelve chunks (K = 576: nine lanes per unit, seven units per wave), three from there on (K = 1536: eight lanes, eight units)
    const int nk = a.K >> 6;
    if (nk > 24) {                                            // K = 3072 (the RWKV7 channel-mix value projection): sixteen lanes per unit, 48 row elements per lane
        if (EPI != GV_PLAIN || NORM) return hipErrorInvalidValue;
        if (a.M <= 1) return gemv_launch_one<GV_PLAIN, false, 1, 3, 0, 48>(st, a);
        if (a.M <= 2) return gemv_launch_one<GV_PLAIN, false, 2, 3, 0, 48>(st, a);
        return gemv_launch_one<GV_PLAIN, false, 4, 3, 0, 48>(st, a);
    }
    const bool three = nk > 12;
    if (a.M <= 1) return three ? gemv_launch_one<EPI, NORM, 1, 3>(st, a) : gemv_launch_one<EPI, NORM, 1, 1>(st, a);
    if (a.M <= 2) return three ? gemv_launch_one<EPI, NORM, 2, 3>(st, a) : gemv_launch_one<EPI, NORM, 2, 1>(st, a);
    return three ? gemv_launch_one<EPI, NORM, 4, 3>(st, a) : gemv_launch_one<EPI, NORM, 4, 1>(st, a);
}
// C[M][N] (natural) = A . W^T (+ R); normw != nullptr: A is the residual stream and RMSNorm(normw, eps) is applied on the way
hipError_t cz_launch_dec_gemv_plain(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int M, int N, int K,
                                    const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (!cz_dec_gemv_ok(M, K, normw != nullptr)) return hipErrorInvalidValue;
    gemv_args a{}; a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.R = R; a.normw = normw; a.eps = eps; a.nunits = N;
    return normw ? gemv_launch<GV_PLAIN, true>(st, a) : gemv_launch<GV_PLAIN, false>(st, a);
}
// the o-projection of the split decode attention: C[M][N] = att . W^T + R with att = the ordered sum of the attention's chunk partials (gemv_args.part)
hipError_t cz_launch_dec_gemv_attsum(hipStream_t st, const float* part, int part_maxc, int max_nk, const int* pos_dev, const uint16_t* W, float* C, int ldc, const float* R,
                                     int M, int N, int n_heads) {
    if (M <= 0) return hipSuccess;
    const int K = n_heads * 64;
    if (!cz_dec_gemv_ok(M, K, false) || K > 768 || !part || !pos_dev || max_nk > 2048 || part_maxc < (max_nk + 63) / 64) return hipErrorInvalidValue;
    gemv_args a{}; a.W = W; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.R = R; a.nunits = N; a.part = part; a.part_maxc = part_maxc; a.pos_dev = pos_dev;
    const int nk = K >> 6, upw = 64 / nk;
    const size_t smem = (size_t)(nk * GV_CS + 64 + nk) * sizeof(float);
    const dim3 grid((N + upw - 1) / upw, M);                   // one row per block: the partials of a row are 100+ loads per lane
    if (max_nk <= 1024) hipLaunchKernelGGL((dec_gemv_kernel<GV_PLAIN, false, 1, 1, 16>), grid, dim3(64), smem, st, a);
    else hipLaunchKernelGGL((dec_gemv_kernel<GV_PLAIN, false, 1, 1, 32>), grid, dim3(64), smem, st, a);
    return hipGetLastError();
}
// fused QKV projection (+ RMSNorm) + RoPE; q -> Q [M][n_heads*64] natural, k / v -> per-stream cache rows at position *pos_dev
hipError_t cz_launch_dec_gemv_qkv(hipStream_t st, const float* A, int lda, const uint16_t* W, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn, const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (!cz_dec_gemv_ok(M, K, normw != nullptr) || !descs || !cs || !sn) return hipErrorInvalidValue;
    gemv_args a{}; a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = (n_heads + 2 * n_kv) * 64; a.K = K; a.C = Q; a.ldc = ldq; a.normw = normw; a.eps = eps;
    a.n_heads = n_heads; a.n_kv = n_kv; a.descs = descs; a.kv_off = kv_off; a.pos_dev = pos_dev; a.cs = cs; a.sn = sn; a.nunits = a.N / 2;
    return normw ? gemv_launch<GV_ROPE, true>(st, a) : gemv_launch<GV_ROPE, false>(st, a);
}
// fused gate|up projection (+ RMSNorm) + SwiGLU; H [M][f] natural; W = [gate rows | up rows]
hipError_t cz_launch_dec_gemv_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* W, float* H, int ldh, int M, int f, int K, const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (!cz_dec_gemv_ok(M, K, normw != nullptr)) return hipErrorInvalidValue;
    gemv_args a{}; a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = 2 * f; a.K = K; a.C = H; a.ldc = ldh; a.normw = normw; a.eps = eps; a.nunits = f;
    return normw ? gemv_launch<GV_SWIGLU, true>(st, a) : gemv_launch<GV_SWIGLU, false>(st, a);
}
// C[M][N] = relu(A . W^T)^2
hipError_t cz_launch_dec_gemv_relusq(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K) {
    if (M <= 0) return hipSuccess;
    if (!cz_dec_gemv_ok(M, K, false) || K > 1536) return hipErrorInvalidValue;
    gemv_args a{}; a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.nunits = N;
    return gemv_launch<GV_RELUSQ, false>(st, a);
}
// rows with row_mask == 0 keep what C held
hipError_t cz_launch_dec_gemv_masked(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K, const unsigned char* row_mask) {
    if (M <= 0) return hipSuccess;
    if (!cz_dec_gemv_ok(M, K, false)) return hipErrorInvalidValue;
    gemv_args a{}; a.A = A; a.lda = lda; a.W = W; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.nunits = N; a.row_mask = row_mask;
    return gemv_launch<GV_PLAIN, false>(st, a);
}
// nprob plain projections C_i[M][N_i] = A_i . W_i^T sharing M (K_i <= 768, W_i rows ldw_i apart), one launch
hipError_t cz_launch_dec_gemv_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* W, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M) {
    if (M <= 0 || nprob <= 0) return hipSuccess;
    if (nprob > GV_GROUP_MAX) return hipErrorInvalidValue;
    gemv_group g{};
    int gx = 0, nkmax = 0;
    for (int i = 0; i < nprob; ++i) {
        if (!cz_dec_gemv_ok(M, K[i], false) || K[i] > 768) return hipErrorInvalidValue;
        gemv_args& a = g.p[i];
        a.A = A[i]; a.lda = lda[i]; a.W = W[i]; a.M = M; a.N = N[i]; a.K = K[i]; a.C = C[i]; a.ldc = ldc[i]; a.nunits = N[i];
        const int nk = K[i] >> 6, upw = 64 / nk, b = (N[i] + upw - 1) / upw;
        gx = b > gx ? b : gx; nkmax = nk > nkmax ? nk : nkmax;
    }
    const int MB = M <= 1 ? 1 : (M <= 2 ? 2 : 4);
    const size_t smem = (size_t)(MB * nkmax * GV_CS + 64 * MB + MB * nkmax) * sizeof(float);
    const dim3 grid(gx, (M + MB - 1) / MB, nprob);
    if (MB == 1) hipLaunchKernelGGL((dec_gemv_group_kernel<1>), grid, dim3(64), smem, st, g);
    else if (MB == 2) hipLaunchKernelGGL((dec_gemv_group_kernel<2>), grid, dim3(64), smem, st, g);
    else hipLaunchKernelGGL((dec_gemv_group_kernel<4>), grid, dim3(64), smem, st, g);
    return hipGetLastError();
}

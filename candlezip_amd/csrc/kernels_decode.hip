// kernels_decode.hip — the lock-step decode step (M = #streams <= 64 rows), latency-oriented.
//
// One token per stream per step: every kernel is tiny, so the step is bound by kernel count, load latency and the KV read
// (46 080 B x ctx per stream, f32), not by MFMA rate.  Numerics are the canonical ones: identical bits to
// the big-M kernels and to the oracle.
//
// Layouts private to this path ("k-interleaved", kq): inside every 64-wide K chunk, element k = 4t + g is stored at
// position 16g + t.  The A/B operand of v_mfma_f32_16x16x4_f32 step t in lane (i, g) is element k = 4t + g, so with this
// layout a lane's 16 operands of a chunk are 16 CONSECUTIVE values (4 x 16-B loads for f32 activations, 2 x 16-B for
// bf16 weights) while the fmaf chain still runs over k ascending.  Weights are re-packed once at load (dec_pack_kernel);
// activations that feed a GEMM (xn, att, h) are written in kq by their producers.
//
// dec_gemm_kernel<EPI>: block = 16 rows x (2 column tiles of 16); one wave per K chunk (rounds of NW waves); chunk
//   partials meet in LDS and are summed in chunk order.  Epilogues: plain(+residual), RoPE+KV-cache write for the fused
//   QKV projection (tile pair = columns c and c+32 of one head), SwiGLU for the fused gate|up projection.
// dec_attn_kernel: one block per (stream, kv head): VALU fmaf chains (lane = key for q.k, lane = d for p.v), K/V rows read
//   once and shared by the rep query heads of the group.
#include "cz_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// tuning build (make dec_dbg, tools/dec_phases.py): block (0, 0) / thread 0 accumulates s_memrealtime ticks (10 ns) per phase of the narrow-group
// kernels; every stamp first waits for the wave's outstanding memory operations, so a phase includes the latency of what it requested
#ifdef CZ_DEC_STAMPS
__device__ unsigned long long g_dec_stamp[8][16];
#define DST_BEGIN unsigned long long st_t = 0; if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) st_t = wall_clock64()
#define DST(kind, ph) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = wall_clock64(); \
                                atomicAdd(&g_dec_stamp[kind][ph], t_ - st_t); if ((ph) == 15) atomicAdd(&g_dec_stamp[kind][0], 1ull); st_t = t_; } } while (0)
extern "C" void cz_debug_dec_stamps(unsigned long long* out128, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_dec_stamp), sizeof(unsigned long long) * 128);
    if (reset) { unsigned long long z[128] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_dec_stamp), z, sizeof z); }
}
#else
#define DST_BEGIN do { } while (0)
#define DST(kind, ph) do { } while (0)
#endif

__host__ __device__ __forceinline__ int kq_index(int k) { return (k & ~63) | ((k & 3) << 4) | ((k & 63) >> 2); }

// Wd[n][kq(k)] = W[n][k]
__global__ void dec_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wd, size_t N, int K) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N * (size_t)K; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / K; const int k = (int)(i % K);
        Wd[n * K + kq_index(k)] = W[i];
    }
}
hipError_t cz_launch_dec_pack(hipStream_t st, const uint16_t* W, uint16_t* Wd, size_t N, int K) {
    hipLaunchKernelGGL(dec_pack_kernel, dim3(2048), dim3(256), 0, st, W, Wd, N, K);
    return hipGetLastError();
}

// RMSNorm for the decode path: same arithmetic as rmsnorm_kernel, output in kq layout (feeds dec_gemm_kernel).
// One block per row, one thread per element (d <= 1024); lane 0 of wave c walks chunk c's canonical 64-sum.
__global__ __launch_bounds__(1024) void dec_rmsnorm_kernel(const float* __restrict__ X, int ldx, const uint16_t* __restrict__ w,
                                                           float* __restrict__ Y, int ldy, int M, int d, float eps) {
    __shared__ float part[16];
    __shared__ float mval;
    const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
    const int m = blockIdx.x;
    const float xv = X[(size_t)m * ldx + tid];
    if (lane == 0) {
        float p = 0.0f;
        const f32x4* src = reinterpret_cast<const f32x4*>(X + (size_t)m * ldx + c * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 v = src[k];
            const float s0 = v[0] * v[0], s1 = v[1] * v[1], s2 = v[2] * v[2], s3 = v[3] * v[3];
            p = p + s0; p = p + s1; p = p + s2; p = p + s3;
        }
        part[c] = p;
    }
    __syncthreads();
    if (tid == 0) {
        const int nch = d >> 6;
        float t = part[0];
        for (int k = 1; k < nch; ++k) t = t + part[k];
        mval = sqrtf(t / (float)d + eps);
    }
    __syncthreads();
    Y[(size_t)m * ldy + kq_index(tid)] = xv / mval * cz_bf16_to_f32(w[tid]);
}
hipError_t cz_launch_dec_rmsnorm(hipStream_t st, const float* X, int ldx, const uint16_t* w, float* Y, int ldy, int M, int d, float eps) {
    if (M <= 0) return hipSuccess;
    if (d % 64 || d > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dec_rmsnorm_kernel, dim3(M), dim3(d), 0, st, X, ldx, w, Y, ldy, M, d, eps);
    return hipGetLastError();
}

enum { EPI_PLAIN = 0, EPI_ROPE = 1, EPI_SWIGLU = 2, EPI_RELUSQ = 3 };
struct dec_gemm_args {
    const float* A; int lda;             // kq layout, [M][K]
    const uint16_t* Wd;                  // kq-packed bf16 [N][K]
    int M, N, K;
    float* C; int ldc;                   // EPI_PLAIN: natural layout out (+ residual R, may alias C); EPI_SWIGLU: kq layout out [M][N/2]
    const float* R;
    // NORM: A is the residual stream x in NATURAL layout; the block applies RMSNorm (weights normw, bf16) on the fly
    const uint16_t* normw; float eps;
    const unsigned char* row_mask;       // EPI_PLAIN: rows with mask == 0 are not stored (RWKV literal symbols do not step the model)
    // EPI_ROPE
    int n_heads, n_kv;
    const cz_rope_desc* descs; size_t kv_off; const int* pos_dev; const float* cs; const float* sn;
};

// MT = 16-row tiles per block (1, 2, 4, 8): one load of the weight slab serves MT x 16 rows.
// NP = column-tile PAIRS per block (1, 2): a wave's A operand (its K chunk of the block's rows) then serves 2 x NP column tiles — wide
// lock-step groups (M >= 256) take MT = 2, NP = 2: 32 x 64 outputs per block, half the activation re-reads and twice the MFMA work per
// operand load of the MT x 1 shape, and still M / 32 x N / 64 blocks (216 for the 768 x 576 o-projection).
template <int EPI, int NORM, int MT, int NP>
__device__ __forceinline__ void dec_gemm_body(const dec_gemm_args& a, const int bx, const int by) {
    extern __shared__ __attribute__((aligned(16))) float part[];       // [NW][G][NP][2][16][17], G = min(MT, 2) row tiles per reduction round
    __shared__ float nsq[MT * 16][17];
    __shared__ float nm[MT * 16];
    constexpr int G = MT < 2 ? MT : 2;
    constexpr int WSTRIDE = G * NP * 544;                               // floats of LDS partials per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = by * (MT * 16);
    int n0[NP], n1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int q = bx * NP + p;                   // pair index
        if (EPI == EPI_ROPE) {                       // pair (c, c+32) inside one 64-wide head; v heads: two plain tiles of the same head
            const int head = q >> 1, j = q & 1;
            n0[p] = head * 64 + j * 16; n1[p] = n0[p] + 32;
        } else if (EPI == EPI_SWIGLU) {              // gate tile + matching up tile
            n0[p] = q * 16; n1[p] = a.N / 2 + n0[p];
        } else {
            n0[p] = q * 32; n1[p] = n0[p] + 16;
        }
    }
    const int K = a.K, nch = K >> 6;
    DST_BEGIN;
    // A narrow group's kernel is a chain of dependent memory round trips (HBM-cold weights ~2 us each), not of arithmetic: everything that
    // does not depend on a result is requested HERE, before the norm phase and its barrier — the first round's weight slab, the plain A
    // operand (PFA: where its registers are affordable), the step position and the rotation's cos / sin (epilogue) — so that the chain is one
    // round trip long instead of three or four.  (Loads only: the arithmetic and its order are untouched.)
    unsigned w0[NP][8], w1[NP][8];
    auto load_w = [&](int c, unsigned (&d0)[NP][8], unsigned (&d1)[NP][8]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint4* b0p = reinterpret_cast<const uint4*>(a.Wd + (size_t)(n0[p] + li) * K + c * 64 + lg * 16);
            const uint4* b1p = reinterpret_cast<const uint4*>(a.Wd + (size_t)(n1[p] + li) * K + c * 64 + lg * 16);
            const uint4 b0a = b0p[0], b0b = b0p[1], b1a = b1p[0], b1b = b1p[1];
            d0[p][0] = b0a.x; d0[p][1] = b0a.y; d0[p][2] = b0a.z; d0[p][3] = b0a.w; d0[p][4] = b0b.x; d0[p][5] = b0b.y; d0[p][6] = b0b.z; d0[p][7] = b0b.w;
            d1[p][0] = b1a.x; d1[p][1] = b1a.y; d1[p][2] = b1a.z; d1[p][3] = b1a.w; d1[p][4] = b1b.x; d1[p][5] = b1b.y; d1[p][6] = b1b.z; d1[p][7] = b1b.w;
        }
    };
    constexpr bool HOIST = MT <= 2 && NP == 1;       // (the wide shapes hold too many accumulators for the early operands, and have the blocks to hide the latency)
    const int NWr = (int)(blockDim.x >> 6);
    constexpr bool HOIST2 = HOIST && MT == 1;        // groups of <= 16 streams: round 1's operands too (K = 1536: 24 chunks on 12 waves)
    unsigned w0b[HOIST2 ? NP : 1][8], w1b[HOIST2 ? NP : 1][8];
    if (HOIST && wave < nch) load_w(wave, w0, w1);
    if constexpr (HOIST2) { if (wave + NWr < nch) load_w(wave + NWr, w0b, w1b); }
    constexpr bool PFA = HOIST && NORM == 0;
    constexpr bool PFA2 = PFA && HOIST2;
    f32x4 apf[PFA ? MT : 1][4], apf2[PFA2 ? MT : 1][4];
    if (PFA && wave < nch) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(a.A + (size_t)min(m0 + m * 16 + li, a.M - 1) * a.lda + wave * 64 + lg * 16);
            apf[m][0] = ap[0]; apf[m][1] = ap[1]; apf[m][2] = ap[2]; apf[m][3] = ap[3];
        }
    }
    if (PFA2 && wave + NWr < nch) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(a.A + (size_t)min(m0 + m * 16 + li, a.M - 1) * a.lda + (wave + NWr) * 64 + lg * 16);
            apf2[m][0] = ap[0]; apf2[m][1] = ap[1]; apf2[m][2] = ap[2]; apf2[m][3] = ap[3];
        }
    }
    constexpr bool PFN = HOIST && NORM == 2;                     // the RMSNorm weights of the wave's chunk (16 bf16 per lane) as well: they sat behind the norm phase's barrier
    float nw_pf[PFN ? 16 : 1];
    if (PFN && wave < nch) {
        const uint16_t* wr = a.normw + wave * 64 + lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) nw_pf[PFN ? t : 0] = cz_bf16_to_f32(wr[4 * t]);
    }
    int pos_pf = 0; float c_pf[NP], s_pf[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { c_pf[p] = 0.0f; s_pf[p] = 0.0f; }
    if (EPI == EPI_ROPE && a.pos_dev && tid < 256) {
        pos_pf = *a.pos_dev;                         // (every shape: two scalars per pair)
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if ((n0[p] >> 6) < a.n_heads + a.n_kv) { const int i = (n0[p] & 63) + (tid & 15); c_pf[p] = a.cs[(size_t)pos_pf * 32 + i]; s_pf[p] = a.sn[(size_t)pos_pf * 32 + i]; }
    }
    DST(EPI + 4 * (NORM ? 1 : 0), 1);          // the hoisted operands have arrived
    // NORM == 2 (single round: nch <= NW, MT <= 2): the wave that owns chunk c reads x[rows][chunk c] once, for the sum of squares AND as
    // its A operand.  The canonical chunk sum p = p + x_k * x_k (product rounded, then added, k ascending) is an MFMA chain with B = 1:
    // fma(x_k^2, 1, p) rounds x_k^2 + p once, exactly like the add.  No separate pass over the rows, no serial 64-long chains.
    float xv[NORM == 2 ? MT : 1][16];
    float mv2[NORM == 2 ? MT : 1];
    if (NORM == 2) {
        if (wave < nch) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // rows past M read row M - 1 (their results are never stored): a load inside a per-element `row < M` branch is waited for on
                // the spot, sixteen dependent round trips instead of one
                const int arow = min(m0 + m * 16 + li, a.M - 1);
                const float* xr = a.A + (size_t)arow * a.lda + wave * 64 + lg;      // natural layout: k = 4t + g
                f32x4 sacc = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 16; ++t) xv[m][t] = xr[4 * t];
#pragma unroll
                for (int t = 0; t < 16; ++t) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[m][t] * xv[m][t], 1.0f, sacc, 0, 0, 0);
                if (li == 0) {                               // C/D: row = 4*lg + r (every column holds the same sum)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nsq[m * 16 + 4 * lg + r][wave] = sacc[r];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float t = nsq[m * 16 + li][0];
            for (int k = 1; k < nch; ++k) t = t + nsq[m * 16 + li][k];
            mv2[m] = sqrtf(t / (float)K + a.eps);
        }
    }
    if (NORM == 1) {
        // candle rms_norm: m = sqrt(sum(x*x)/d + eps) with the canonical chunk-64 sum; y = x / m * w   (K <= 1024)
        for (int task = tid; task < MT * 16 * nch; task += blockDim.x) {
            const int r = task / nch, c = task % nch;
            float p = 0.0f;
            if (m0 + r < a.M) {
                const f32x4* src = reinterpret_cast<const f32x4*>(a.A + (size_t)(m0 + r) * a.lda + c * 64);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const f32x4 v = src[k];
                    const float s0 = v[0] * v[0], s1 = v[1] * v[1], s2 = v[2] * v[2], s3 = v[3] * v[3];
                    p = p + s0; p = p + s1; p = p + s2; p = p + s3;
                }
            }
            nsq[r][c] = p;
        }
        __syncthreads();
        if (tid < MT * 16) {
            float t = nsq[tid][0];
            for (int k = 1; k < nch; ++k) t = t + nsq[tid][k];
            nm[tid] = sqrtf(t / (float)K + a.eps);
        }
        __syncthreads();
    }
    DST(EPI + 4 * (NORM ? 1 : 0), 2);          // norm phase done
    float tot0[NP][MT], tot1[NP][MT];            // element (r = tid >> 4, c = tid & 15) of both column tiles of pair p, row tile m; threads 0..255
    float res0[NP][MT], res1[NP][MT];            // residual, fetched now so that its latency hides under the weight loads + MFMA chain
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            tot0[p][m] = 0.0f; tot1[p][m] = 0.0f; res0[p][m] = 0.0f; res1[p][m] = 0.0f;
            if (EPI == EPI_PLAIN && a.R && tid < 256) {
                const int row = min(m0 + m * 16 + (tid >> 4), a.M - 1);          // (clamped, unconditional: all residual loads in flight together)
                res0[p][m] = a.R[(size_t)row * a.ldc + n0[p] + (tid & 15)]; res1[p][m] = a.R[(size_t)row * a.ldc + n1[p] + (tid & 15)];
            }
        }
    for (int c0 = 0; c0 < nch; c0 += NW) {
        const int c = c0 + wave;
        f32x4 acc0[NP][MT], acc1[NP][MT];
        if (c < nch) {
            if (!HOIST || c0 > (HOIST2 ? NW : 0)) load_w(c, w0, w1);              // (HOIST: round 0's slab — HOIST2: round 1's too — was requested at the top of the kernel)
            if constexpr (HOIST2) {
                if (c0 == NW) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { w0[p][e] = w0b[p][e]; w1[p][e] = w1b[p][e]; }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int arow = min(m0 + m * 16 + li, a.M - 1);                  // (clamped: no branch between the row tiles' loads)
                f32x4 av[4];
                if (NORM == 2 && PFN) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) av[t >> 2][t & 3] = xv[m][t] / mv2[m] * nw_pf[PFN ? t : 0];
                } else if (NORM == 2) {
                    const uint16_t* wr = a.normw + c * 64 + lg;
#pragma unroll
                    for (int t = 0; t < 16; ++t) av[t >> 2][t & 3] = xv[m][t] / mv2[m] * cz_bf16_to_f32(wr[4 * t]);
                } else if (NORM) {
                    const float* xr = a.A + (size_t)arow * a.lda + c * 64 + lg;        // natural layout: k = 4t + g
                    const uint16_t* wr = a.normw + c * 64 + lg;
                    const float mv = nm[m * 16 + li];
#pragma unroll
                    for (int t = 0; t < 16; ++t) av[t >> 2][t & 3] = xr[4 * t] / mv * cz_bf16_to_f32(wr[4 * t]);
                } else if (PFA && c0 == 0) {
                    av[0] = apf[PFA ? m : 0][0]; av[1] = apf[PFA ? m : 0][1]; av[2] = apf[PFA ? m : 0][2]; av[3] = apf[PFA ? m : 0][3];
                } else if (PFA2 && c0 == NW) {
                    av[0] = apf2[PFA2 ? m : 0][0]; av[1] = apf2[PFA2 ? m : 0][1]; av[2] = apf2[PFA2 ? m : 0][2]; av[3] = apf2[PFA2 ? m : 0][3];
                } else {
                    const f32x4* ap = reinterpret_cast<const f32x4*>(a.A + (size_t)arow * a.lda + c * 64 + lg * 16);
                    av[0] = ap[0]; av[1] = ap[1]; av[2] = ap[2]; av[3] = ap[3];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x4 x0 = f32x4{0, 0, 0, 0}, x1 = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float av_t = av[t >> 2][t & 3];
                        const float bv0 = (t & 1) ? cz_bits_f32(w0[p][t >> 1] & 0xffff0000u) : cz_bits_f32(w0[p][t >> 1] << 16);
                        const float bv1 = (t & 1) ? cz_bits_f32(w1[p][t >> 1] & 0xffff0000u) : cz_bits_f32(w1[p][t >> 1] << 16);
                        x0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av_t, bv0, x0, 0, 0, 0);
                        x1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av_t, bv1, x1, 0, 0, 0);
                    }
                    acc0[p][m] = x0; acc1[p][m] = x1;
                }
            }
        }
        DST(EPI + 4 * (NORM ? 1 : 0), 3);      // MFMA chains of the round
        // ordered reduction of the chunk partials, G row tiles per round
#pragma unroll
        for (int mg = 0; mg < MT; mg += G) {
            if (c < nch) {
                float* pw = part + wave * WSTRIDE;
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {       // C/D: col = li, row = 4*lg + r
                            pw[((q * NP + p) * 2 + 0) * 272 + (4 * lg + r) * 17 + li] = acc0[p][mg + q][r];
                            pw[((q * NP + p) * 2 + 1) * 272 + (4 * lg + r) * 17 + li] = acc1[p][mg + q][r];
                        }
            }
            __syncthreads();
            if (tid < 256) {
                const int r = tid >> 4, cc = tid & 15;
                const int lim = nch - c0 < NW ? nch - c0 : NW;
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        for (int w = 0; w < lim; ++w) {
                            const float p0 = part[w * WSTRIDE + ((q * NP + p) * 2 + 0) * 272 + r * 17 + cc];
                            const float p1 = part[w * WSTRIDE + ((q * NP + p) * 2 + 1) * 272 + r * 17 + cc];
                            if (c0 + w == 0) { tot0[p][mg + q] = p0; tot1[p][mg + q] = p1; } else { tot0[p][mg + q] = tot0[p][mg + q] + p0; tot1[p][mg + q] = tot1[p][mg + q] + p1; }
                        }
            }
            __syncthreads();
        }
        DST(EPI + 4 * (NORM ? 1 : 0), 4);      // ordered reduction of the round
    }
    if (tid >= 256) return;
    const int r = tid >> 4, cc = tid & 15;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m0 + m * 16 + r;
        if (row >= a.M) continue;
        const float t0 = tot0[p][m], t1 = tot1[p][m];
        if (EPI == EPI_RELUSQ) {                                       // RWKV7 channel-mix key: relu(x)^2, kq layout (feeds the value projection)
            const float z0 = t0 > 0.0f ? t0 : 0.0f, z1 = t1 > 0.0f ? t1 : 0.0f;
            a.C[(size_t)row * a.ldc + kq_index(n0[p] + cc)] = z0 * z0;
            a.C[(size_t)row * a.ldc + kq_index(n1[p] + cc)] = z1 * z1;
        } else if (EPI == EPI_PLAIN) {
            if (a.row_mask && !a.row_mask[row]) continue;
            float v0 = t0, v1 = t1;
            if (a.R) { v0 = v0 + res0[p][m]; v1 = v1 + res1[p][m]; }
            a.C[(size_t)row * a.ldc + n0[p] + cc] = v0;
            a.C[(size_t)row * a.ldc + n1[p] + cc] = v1;
        } else if (EPI == EPI_SWIGLU) {
            a.C[(size_t)row * a.ldc + kq_index(n0[p] + cc)] = cz_siluf(t0) * t1;      // silu(gate) * up  (llama.rs MLP)
        } else {
            cz_rope_desc d = a.descs[row];           // one row per stream
            const int pos = a.pos_dev ? pos_pf : d.pos0;
            const int head = n0[p] >> 6, i = (n0[p] & 63) + cc;                       // rotation index i in [0, 32)
            if (head < a.n_heads + a.n_kv) {
                const float c_ = a.pos_dev ? c_pf[p] : a.cs[(size_t)pos * 32 + i], s_ = a.pos_dev ? s_pf[p] : a.sn[(size_t)pos * 32 + i];
                const float aa = t0 * c_, bb = t1 * s_, c2 = t0 * s_, dd = t1 * c_;      // candle rope: products rounded separately
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
    DST(EPI + 4 * (NORM ? 1 : 0), 15);         // epilogue
}

template <int EPI, int NORM, int MT, int NP>
__global__ __launch_bounds__(768) void dec_gemm_kernel(dec_gemm_args a) { dec_gemm_body<EPI, NORM, MT, NP>(a, blockIdx.x, blockIdx.y); }

// Several independent plain GEMMs of one decode step in ONE launch (blockIdx.z = problem): the RWKV7 time-mix needs seven
// projections of differently mixed inputs, then four LoRA up-projections; as separate launches they cost ~7 us each for a few
// hundred ns of work.  Blocks past a problem's own column count leave at once (uniformly, before any barrier).
#define DEC_GROUP_MAX 8
struct dec_gemm_group { dec_gemm_args a[DEC_GROUP_MAX]; int gx[DEC_GROUP_MAX]; };
template <int MT>
__global__ __launch_bounds__(768) void dec_gemm_group_kernel(dec_gemm_group g) {
    const int z = blockIdx.z;
    if ((int)blockIdx.x >= g.gx[z]) return;
    dec_gemm_body<EPI_PLAIN, 0, MT, 1>(g.a[z], blockIdx.x, blockIdx.y);
}

int g_dec_nw_cap = 12;           // tuning knob (cz_debug_set "dec_nw_cap"): waves per decode-GEMM block (each wave owns K chunks round-robin)
static int dec_nw(int K) { int nch = K / 64; nch = nch < g_dec_nw_cap ? nch : g_dec_nw_cap; return nch < 4 ? 4 : nch; }   // >= 256 threads: one per element of a tile
template <int EPI, int NORM, int MT, int NP = 1>
static hipError_t dec_gemm_launch_mt(hipStream_t st, const dec_gemm_args& a, int gx) {
    const int NW = dec_nw(a.K);
    const int G = MT < 2 ? MT : 2;
    const size_t smem = (size_t)NW * G * NP * 2 * 16 * 17 * sizeof(float);
    auto kern = dec_gemm_kernel<EPI, NORM, MT, NP>;          // (NP = 2 blocks pass 48 KB of LDS: cz_dec_gemm_prepare, once per device at session open)
    hipLaunchKernelGGL(kern, dim3(gx / NP, (a.M + MT * 16 - 1) / (MT * 16)), dim3(64 * NW), smem, st, a);
    return hipGetLastError();
}
template <int EPI, int NORM, int MT, int NP>
static hipError_t dec_gemm_prepare_one() {     // up to 12 waves x min(MT, 2) row tiles x NP pairs x 2 tiles of [16][17] partials: past 48 KB from MT = 2 on
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm_kernel<EPI, NORM, MT, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 2 * NP * 2 * 16 * 17 * (int)sizeof(float));
}
template <int EPI>
static hipError_t dec_gemm_prepare_epi() {
    hipError_t e = dec_gemm_prepare_one<EPI, 0, 2, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 0, 4, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 0, 8, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 1, 2, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 1, 4, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 1, 8, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 2, 2, 1>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 0, 2, 2>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 1, 2, 2>();
    if (e == hipSuccess) e = dec_gemm_prepare_one<EPI, 2, 2, 2>();
    return e;
}
hipError_t cz_dec_gemm_prepare() {
    hipError_t e = dec_gemm_prepare_epi<EPI_PLAIN>();
    if (e == hipSuccess) e = dec_gemm_prepare_epi<EPI_ROPE>();
    if (e == hipSuccess) e = dec_gemm_prepare_epi<EPI_SWIGLU>();
    if (e == hipSuccess) e = dec_gemm_prepare_epi<EPI_RELUSQ>();
    return e;
}
int g_dec_grid_target = 384;     // tuning knob (cz_debug_set "dec_grid_target")
int g_dec_mt_max = 8;            // tuning knob (cz_debug_set "dec_mt_max"): most 16-row tiles per block
template <int EPI, int NORM>
static hipError_t dec_gemm_launch_n(hipStream_t st, const dec_gemm_args& a, int gx) {
    // row tiles per block: keep the grid near one wave of blocks (<= ~384) while it still has >= 64 of them
    const int mt_all = (a.M + 15) / 16;
    int mt = 1;
    while (mt < g_dec_mt_max && mt < mt_all && (long)gx * ((mt_all + mt - 1) / mt) > g_dec_grid_target) mt *= 2;
    switch (mt) {
        case 1: return dec_gemm_launch_mt<EPI, NORM, 1>(st, a, gx);
        case 2: return dec_gemm_launch_mt<EPI, NORM, 2>(st, a, gx);
        case 4: return dec_gemm_launch_mt<EPI, NORM, 4>(st, a, gx);
        default: return dec_gemm_launch_mt<EPI, NORM, 8>(st, a, gx);
    }
}
int g_dec_norm_mfma = 1;         // tuning knob (cz_debug_set "dec_norm_mfma"): 0 = the separate sum-of-squares pass (NORM = 1) everywhere
int g_dec_wide_rows = 256;       // tuning knob (cz_debug_set "dec_wide_rows"): from this many rows on, blocks of 32 rows x 64 columns (MT = 2, NP = 2)
template <int EPI>
static hipError_t dec_gemm_launch(hipStream_t st, const dec_gemm_args& a, int gx) {
    if (a.M >= g_dec_wide_rows && gx % 2 == 0) {
        if (!a.normw) return dec_gemm_launch_mt<EPI, 0, 2, 2>(st, a, gx);
        if (a.K > 1024) return hipErrorInvalidValue;
        if (g_dec_norm_mfma && a.K / 64 <= dec_nw(a.K)) return dec_gemm_launch_mt<EPI, 2, 2, 2>(st, a, gx);
        return dec_gemm_launch_mt<EPI, 1, 2, 2>(st, a, gx);
    }
    if (a.normw) {
        if (a.K > 1024) return hipErrorInvalidValue;
        if (g_dec_norm_mfma && a.K / 64 <= dec_nw(a.K)) {       // one round of chunks: the fused MFMA sum of squares (at most two row tiles per block)
            const int mt_all = (a.M + 15) / 16;
            if (mt_all >= 2 && (long)gx * mt_all > g_dec_grid_target) return dec_gemm_launch_mt<EPI, 2, 2>(st, a, gx);
            return dec_gemm_launch_mt<EPI, 2, 1>(st, a, gx);
        }
        return dec_gemm_launch_n<EPI, 1>(st, a, gx);
    }
    return dec_gemm_launch_n<EPI, 0>(st, a, gx);
}
// C[M][N] (natural) = A(kq) . W^T (+ R)
hipError_t cz_launch_dec_gemm_plain(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, const float* R, int M, int N, int K,
                                    const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (N % 32 || K % 64) return hipErrorInvalidValue;
    dec_gemm_args a{}; a.A = A; a.lda = lda; a.Wd = Wd; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.R = R; a.normw = normw; a.eps = eps;
    return dec_gemm_launch<EPI_PLAIN>(st, a, N / 32);
}
hipError_t cz_launch_dec_gemm_masked(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, const float* R, int M, int N, int K,
                                     const unsigned char* row_mask) {
    if (M <= 0) return hipSuccess;
    if (N % 32 || K % 64) return hipErrorInvalidValue;
    dec_gemm_args a{}; a.A = A; a.lda = lda; a.Wd = Wd; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc; a.R = R; a.row_mask = row_mask;
    return dec_gemm_launch<EPI_PLAIN>(st, a, N / 32);
}
// nprob plain GEMMs C_i[M][N_i] = A_i(kq) . W_i^T sharing M, one launch
hipError_t cz_launch_dec_gemm_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* Wd, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M) {
    if (M <= 0 || nprob <= 0) return hipSuccess;
    if (nprob > DEC_GROUP_MAX) return hipErrorInvalidValue;
    dec_gemm_group g{};
    int gxmax = 0, kmax = 0;
    for (int i = 0; i < nprob; ++i) {
        if (N[i] % 32 || K[i] % 64) return hipErrorInvalidValue;
        dec_gemm_args& a = g.a[i];
        a.A = A[i]; a.lda = lda[i]; a.Wd = Wd[i]; a.M = M; a.N = N[i]; a.K = K[i]; a.C = C[i]; a.ldc = ldc[i];
        g.gx[i] = N[i] / 32;
        gxmax = gxmax > g.gx[i] ? gxmax : g.gx[i]; kmax = kmax > K[i] ? kmax : K[i];
    }
    const int NW = dec_nw(kmax);
    const size_t smem = (size_t)NW * 1 * 2 * 16 * 17 * sizeof(float);
    hipLaunchKernelGGL((dec_gemm_group_kernel<1>), dim3(gxmax, (M + 15) / 16, nprob), dim3(64 * NW), smem, st, g);
    return hipGetLastError();
}
// C (kq layout, [M][N]) = relu(A(kq) . W^T)^2
hipError_t cz_launch_dec_gemm_relusq(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* C, int ldc, int M, int N, int K) {
    if (M <= 0) return hipSuccess;
    if (N % 32 || K % 64) return hipErrorInvalidValue;
    dec_gemm_args a{}; a.A = A; a.lda = lda; a.Wd = Wd; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
    return dec_gemm_launch<EPI_RELUSQ>(st, a, N / 32);
}
// fused QKV projection + RoPE; q -> Q [M][n_heads*64] natural, k/v -> per-stream cache rows at position *pos_dev
hipError_t cz_launch_dec_gemm_qkv(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn,
                                  const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (K % 64) return hipErrorInvalidValue;
    dec_gemm_args a{}; a.normw = normw; a.eps = eps; a.A = A; a.lda = lda; a.Wd = Wd; a.M = M; a.N = (n_heads + 2 * n_kv) * 64; a.K = K; a.C = Q; a.ldc = ldq;
    a.n_heads = n_heads; a.n_kv = n_kv; a.descs = descs; a.kv_off = kv_off; a.pos_dev = pos_dev; a.cs = cs; a.sn = sn;
    return dec_gemm_launch<EPI_ROPE>(st, a, (n_heads + 2 * n_kv) * 2);
}
// true when a norm-fused projection of K inputs takes the single-round MFMA sum-of-squares path (otherwise every column block would
// repeat a pass over its rows, and wide projections normalise once with dec_rmsnorm_kernel instead)
bool cz_dec_gemm_norm_is_cheap(int K) { return g_dec_norm_mfma && K % 64 == 0 && K <= 1024 && K / 64 <= dec_nw(K); }
// fused gate|up projection + SwiGLU; H [M][f] in kq layout
hipError_t cz_launch_dec_gemm_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wd, float* H, int ldh, int M, int f, int K,
                                     const uint16_t* normw, float eps) {
    if (M <= 0) return hipSuccess;
    if (f % 64 || K % 64) return hipErrorInvalidValue;
    dec_gemm_args a{}; a.normw = normw; a.eps = eps; a.A = A; a.lda = lda; a.Wd = Wd; a.M = M; a.N = 2 * f; a.K = K; a.C = H; a.ldc = ldh;
    return dec_gemm_launch<EPI_SWIGLU>(st, a, f / 16);
}

// ------------------------------------------------------------------------------------------------------------
// decode attention: block = (stream, kv head), 512 threads (8 waves).  Same arithmetic as attention_kernel.
// Wave w owns key chunks w, w+8, ... through all phases (scores stay in registers), so only three block barriers are
// needed: after the local maxima, after the chunk sums, after the chunk partials of P.V.
// HBM-bound: algorithmic bytes = 2 * nk * 256 B per block (K and V rows of the group's kv head, f32).
#define DA_MAXREP 4
#define DA_MAXW 16                                // most waves per block (LDS sizing of the per-wave maxima)
typedef __attribute__((address_space(1))) const float gfloat;     // K/V pointers come out of a descriptor in memory: tell the
                                                                  // compiler they are global (global_load, not flat_load)
// rep = query heads per kv head (compile-time so that the per-head code carries no branches); DA_WAVES = 8 waves per block for whole-device
// grids, 16 where the grid is small (a single stream has n_kv blocks in all: twice the waves halve the chunks each wave walks serially)
// DA_MAXC = key chunks per wave: DA_WAVES * DA_MAXC chunks = the longest context of the instantiation (32 chunks = 2048 keys for the two
// everyday shapes; 16 waves x 8 = 128 chunks = 8192 keys = max_position_embeddings of SmolLM2 for the long variant, whose chunk partials
// of P.V are handed over in rounds of DA_WAVES chunks so that its LDS stays under 160 KB: scores 98 KB at rep 3).
#ifndef CZ_DA_VPRE
#define CZ_DA_VPRE 1         // the whole-device form requests its first V round before the softmax phases (tuning build `make da_vpre0`: 0 = not)
#endif
#ifndef CZ_DA_VPRE_KEYS
#define CZ_DA_VPRE_KEYS 16   // keys of that first V round (tuning build `make da_vpre32`: 32)
#endif
#ifndef CZ_DA_OCC
#define CZ_DA_OCC 4          // waves per SIMD the whole-device form is compiled for (tuning build `make da_occ6`: 6 = three blocks per CU, 80 VGPRs)
#endif
template <int rep, int DA_WAVES, int DA_MAXC>
__global__ __launch_bounds__(DA_WAVES * 64, (DA_WAVES * DA_MAXC > 32 ? 1 : (DA_WAVES <= 8 ? CZ_DA_OCC : 4))) void dec_attn_kernel(const float* __restrict__ Q, int ldq, float* __restrict__ O, int ldo,
                                                              const cz_seq_desc* __restrict__ descs, int n_heads, int n_kv, int s_stride,
                                                              size_t kv_off, const int* __restrict__ pos_dev, int out_kq, int kv_div) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cz_seq_desc sd = descs[blockIdx.x];
    const int pos = pos_dev ? *pos_dev : sd.pos0;
    const int nk = pos + 1, nch = (nk + 63) >> 6;
    // kv_div == 1: block = (stream, kv head), its `rep` query heads share the K/V it streams.  kv_div > 1 (rep == 1; small grids, e.g. ONE
    // stream): block = (stream, QUERY head), kv head = blockIdx.y / kv_div — three times the blocks re-read the same K/V, each a third of the work
    const int kvh = blockIdx.y / kv_div, qh0 = kv_div > 1 ? (int)blockIdx.y : kvh * rep;
    gfloat* Kb = (gfloat*)(sd.k + kv_off + (sd.kt_stride ? (size_t)kvh * 64 * sd.kt_stride : (size_t)kvh * 64));
    gfloat* Vb = (gfloat*)(sd.v + kv_off + kvh * 64);
    // LDS sized by the launch for `rep` heads and the longest context the caller will reach (s_stride = 64 * chunks + 4): 26 KB at 1024
    // keys, rep 3 — small enough that a GEMM block of another decode lane fits on the CU beside two attention blocks
    const int nch_max = (s_stride - 4) >> 6;
    float* qs = smem;                                   // [rep][64]
    float* S = qs + rep * 64;                           // [rep][s_stride]  e, then p
    constexpr bool DEEP = DA_WAVES == 16 && DA_MAXC == 2 && rep == 1;   // small grids, one block per QUERY head: every load of a chunk in flight at once (see phase 1; rep > 1 would spill under the 128-register cap of 16 waves)
    constexpr bool LONG = DA_WAVES * DA_MAXC > 32;       // contexts past 2048 keys (rounds of DA_WAVES chunk partials instead of all nch_max at once)
    constexpr int CS = DA_WAVES * DA_MAXC;               // chunks per block = stride of the chunk sums
    float* part = S + rep * s_stride;                   // [nch_max][rep][64]   chunk partials of P.V ([DA_WAVES][rep][64] per round when LONG)
    float* red = part + (LONG ? DA_WAVES : nch_max) * rep * 64;             // [DA_WAVES][DA_MAXREP] local maxima
    float* csum = red + DA_MAXW * DA_MAXREP;            // [rep][CS]       chunk sums of e
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DST_BEGIN;
    if (tid < rep * 64) qs[tid] = Q[(size_t)sd.row0 * ldq + (qh0 + (tid >> 6)) * 64 + (tid & 63)];
    __syncthreads();
    DST(3, 1);                                  // descriptor, position, Q in LDS
    // ---- phase 1: scores of the wave's chunks (lane = key), kept in registers
    float sc[DA_MAXC][DA_MAXREP];
    float lm[DA_MAXREP] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < DA_MAXC; ++i) {
        const int c = wave + DA_WAVES * i;
#pragma unroll
        for (int h = 0; h < DA_MAXREP; ++h) sc[i][h] = -INFINITY;
        if (c < nch) {
            const int j = c * 64 + lane;
            if (j < nk) {
                float s[DA_MAXREP] = {0.0f, 0.0f, 0.0f, 0.0f};
                // transposed cache: for every d the 64 lanes (keys) read 256 contiguous bytes.  Four real (not unrolled) rounds of 16 d keep
                // the kernel under 128 VGPRs -> two blocks per CU: one block's softmax phase overlaps the other's loads, and the 384 blocks
                // of a 128-stream step are resident at once instead of running in one and a half rounds
                auto k_round = [&](const int g) {
                    float kreg[16];
                    if (sd.kt_stride) {
#pragma unroll
                        for (int dd = 0; dd < 16; ++dd) kreg[dd] = __builtin_nontemporal_load(Kb + (size_t)(g * 16 + dd) * sd.kt_stride + j);   // streamed once per step: keep it out of the way of the weights
                    } else {
                        const __attribute__((address_space(1))) f32x4* kr = (const __attribute__((address_space(1))) f32x4*)(Kb + (size_t)j * sd.kv_stride + g * 16);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) { const f32x4 t4 = kr[q4]; kreg[4 * q4] = t4[0]; kreg[4 * q4 + 1] = t4[1]; kreg[4 * q4 + 2] = t4[2]; kreg[4 * q4 + 3] = t4[3]; }
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 kv = f32x4{kreg[4 * q4], kreg[4 * q4 + 1], kreg[4 * q4 + 2], kreg[4 * q4 + 3]};
#pragma unroll
                        for (int h = 0; h < DA_MAXREP; ++h)
                            if (h < rep) {
                                const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + h * 64 + g * 16 + q4 * 4);
                                s[h] = cz_fma(qv[0], kv[0], s[h]); s[h] = cz_fma(qv[1], kv[1], s[h]);
                                s[h] = cz_fma(qv[2], kv[2], s[h]); s[h] = cz_fma(qv[3], kv[3], s[h]);
                            }
                    }
                                };
                if (DEEP) {          // the 16-wave shape of small grids (e.g. ONE stream): a block is alone on its CU, four rounds = four dependent round trips;
#pragma unroll                       // fully unrolled, the chunk's 64 loads are in flight at once
                    for (int g = 0; g < 4; ++g) k_round(g);
                } else {
#pragma unroll 1
                    for (int g = 0; g < 4; ++g) k_round(g);
                }
#pragma unroll
                for (int h = 0; h < DA_MAXREP; ++h) if (h < rep) { sc[i][h] = s[h] * 0.125f; lm[h] = sc[i][h] > lm[h] ? sc[i][h] : lm[h]; }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < DA_MAXREP; ++h) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float y = __shfl_xor(lm[h], o); lm[h] = y > lm[h] ? y : lm[h]; }
    }
    DST(3, 2);                                  // phase 1: K streamed, scores
    // DEEP: the V slab of the wave's first chunk is requested NOW — the K registers are free — and arrives under the two softmax phases and their
    // barriers (the bit pattern of what is loaded does not depend on them); phase 3 then starts on resident data
    // The whole-device form (two blocks per CU) requests the first 16-key round of that slab: enough to keep loads in flight through the softmax
    // phases, whose barriers otherwise leave this block's share of the memory pipe empty (CZ_DA_VPRE = 0: off)
    constexpr int VPRE = DEEP ? 64 : ((!LONG && DA_WAVES <= 8 && CZ_DA_VPRE) ? CZ_DA_VPRE_KEYS : 0);
    float vpre[VPRE ? VPRE : 1];
    if (VPRE) {
        const int c = wave;
        if (c < nch) {
            const int jend = nk - c * 64 < 64 ? nk - c * 64 : 64;
            gfloat* vp = Vb + (size_t)(c * 64) * sd.kv_stride + lane;
#pragma unroll
            for (int jx = 0; jx < (VPRE ? VPRE : 1); ++jx) { const int jj = jx < jend ? jx : jend - 1; vpre[VPRE ? jx : 0] = __builtin_nontemporal_load(vp + (size_t)jj * sd.kv_stride); }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < DA_MAXREP; ++h) red[wave * DA_MAXREP + h] = lm[h];
    }
    __syncthreads();
    DST(3, 3);                                  // barrier after the local maxima
    // ---- phase 2: e = exp(s - max), chunk sums in the canonical order (one lane per head walks the 64 values)
    float gm[DA_MAXREP];
#pragma unroll
    for (int h = 0; h < DA_MAXREP; ++h) {
        float m = red[h];
#pragma unroll
        for (int w = 1; w < DA_WAVES; ++w) { const float y = red[w * DA_MAXREP + h]; m = y > m ? y : m; }
        gm[h] = m;
    }
#pragma unroll
    for (int i = 0; i < DA_MAXC; ++i) {
        const int c = wave + DA_WAVES * i;
        if (c < nch) {
#pragma unroll
            for (int h = 0; h < DA_MAXREP; ++h)
                if (h < rep) { sc[i][h] = cz_expf(sc[i][h] - gm[h]); S[h * s_stride + c * 64 + lane] = sc[i][h]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // this wave's LDS writes are visible to its own lanes
    {   // one lane per (chunk of this wave, head): the wave's DA_MAXC x rep chains of 64 adds run side by side (they used to run one after the
        // other on `rep` lanes: 4 x 64 dependent adds between the two barriers of the softmax, with no load of this block in flight)
        const int i = lane / DA_MAXREP, h = lane % DA_MAXREP, c = wave + DA_WAVES * i;
        if (lane < DA_MAXC * DA_MAXREP && h < rep && c < nch) {
            const float* p = S + h * s_stride + c * 64;
            float acc = 0.0f;
#pragma unroll 16
            for (int k = 0; k < 64; ++k) acc = acc + p[k];
            csum[h * CS + c] = acc;
        }
    }
    __syncthreads();
    DST(3, 4);                                  // phase 2: exp, chunk sums, barrier
    // ---- phase 3: total (chunks left to right), p = e / total, P.V chunk partials (lane = d, keys ascending)
    float tot[DA_MAXREP];
#pragma unroll
    for (int h = 0; h < DA_MAXREP; ++h) {
        float t = 0.0f;
        if (h < rep) { t = csum[h * CS]; for (int c = 1; c < nch; ++c) t = t + csum[h * CS + c]; }
        tot[h] = t;
    }
#pragma unroll
    for (int i = 0; i < DA_MAXC; ++i) {
        const int c = wave + DA_WAVES * i;
        if (c < nch) {
#pragma unroll
            for (int h = 0; h < DA_MAXREP; ++h) if (h < rep) S[h * s_stride + c * 64 + lane] = sc[i][h] / tot[h];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float osum = 0.0f;                                   // LONG: thread (h, d)'s running total over the chunks, in chunk order
#pragma unroll
    for (int i = 0; i < DA_MAXC; ++i) {
        const int c = wave + DA_WAVES * i;
        if (LONG && DA_WAVES * i >= nch) break;          // (block-uniform: the rounds below hold barriers)
        if (c < nch) {
            float acc[DA_MAXREP] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int jend = nk - c * 64 < 64 ? nk - c * 64 : 64;
            gfloat* vp = Vb + (size_t)(c * 64) * sd.kv_stride + lane;
            // the 64-key slab of this lane's column in four real rounds of 16 keys (register budget, see phase 1).  Unconditional loads (row
            // index clamped) + select afterwards: a per-element "load or 0" on a runtime bound makes hipcc branch around every load and wait
            // for each one.  Keys past nk have p == 0 (exp(-inf)) and v == 0: fma(0, 0, acc) == acc, so the chain needs no guard.
            auto v_round = [&](const int g) {
                float vreg[16];
                if (VPRE && i == 0 && g * 16 < VPRE) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) vreg[j] = vpre[VPRE ? (g * 16 + j) % (VPRE ? VPRE : 1) : 0];
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const int jx = g * 16 + j; const int jj = jx < jend ? jx : jend - 1; vreg[j] = __builtin_nontemporal_load(vp + (size_t)jj * sd.kv_stride); }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) vreg[j] = g * 16 + j < jend ? vreg[j] : 0.0f;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    f32x4 pv[DA_MAXREP];
#pragma unroll
                    for (int h = 0; h < DA_MAXREP; ++h) if (h < rep) pv[h] = *reinterpret_cast<const f32x4*>(S + h * s_stride + c * 64 + g * 16 + 4 * j4);   // 4 keys per LDS read
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int h = 0; h < DA_MAXREP; ++h) if (h < rep) acc[h] = cz_fma(pv[h][u], vreg[4 * j4 + u], acc[h]);
                }
                        };
            if (DEEP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v_round(g);
            } else {
#pragma unroll 1
                for (int g = 0; g < 4; ++g) v_round(g);
            }
#pragma unroll
            for (int h = 0; h < DA_MAXREP; ++h) if (h < rep) part[((LONG ? wave : c) * rep + h) * 64 + lane] = acc[h];
        }
        if (LONG) {     // round i = chunks [DA_WAVES i, DA_WAVES (i + 1)): a contiguous range, so adding the rounds one after the other keeps the chunk order
            __syncthreads();
            if (tid < rep * 64) {
                const int h = tid >> 6, dd = tid & 63;
                const int lim = nch - DA_WAVES * i < DA_WAVES ? nch - DA_WAVES * i : DA_WAVES;
                for (int w = 0; w < lim; ++w) { const float pw = part[(w * rep + h) * 64 + dd]; osum = (i == 0 && w == 0) ? pw : osum + pw; }
            }
            __syncthreads();
        }
    }
    DST(3, 5);                                  // phase 3: V streamed, P.V partials
    if (!LONG) __syncthreads();
    if (tid < rep * 64) {
        const int h = tid >> 6, dd = tid & 63;
        float t = osum;
        if (!LONG) {
            t = part[(0 * rep + h) * 64 + dd];
            for (int c = 1; c < nch; ++c) t = t + part[(c * rep + h) * 64 + dd];
        }
        const int col = (qh0 + h) * 64 + dd;
        O[(size_t)sd.row0 * ldo + (out_kq ? kq_index(col) : col)] = t;          // kq layout feeds dec_gemm_kernel's o-projection, natural layout the tiled GEMM
    }
    DST(3, 15);                                 // ordered sum of the partials, output
}
// ------------------------------------------------------------------------------------------------------------
// Split decode attention for a HANDFUL of streams (nstreams x n_heads <= dec_attn_split_max blocks of the one-kernel form).  There a query head's
// block streams its whole K and V alone on one CU — 393 KB at 767 keys, i.e. at the ~65 GB/s ONE CU streams at, 6 of the kernel's 13 us
// (profiles/r03_dec_phases_b1.json).  Spread over (head, 64-key chunk) blocks the same bytes move on 12x as many CUs.  The exact softmax needs the
// maximum over ALL keys before any exponential, so this is three dependent launches instead of one (2-3 us each):
//   da_split_scores_kernel : block = (stream, q head, chunk): the chunk's 64 scores (the same d-ascending fmaf chain) and its maximum
//   da_split_pv_kernel     : block = (stream, q head, chunk): global max of the head's chunk maxima; e = exp(s - max) of EVERY chunk (3 KB of
//                            scores: cheap to redo per block), the canonical chunk sums (lane c walks chunk c), total left to right,
//                            p = e / total of the own chunk, P.V partial of the own chunk (lane = d, keys ascending)
//   da_split_sum_kernel    : block = (stream, q head): the chunk partials added left to right -> O
// Arithmetic and order are dec_attn_kernel's, statement for statement: the same bits.
#define DA_SPLIT_MAXC 32                               // chunks of 64 keys: contexts up to 2048
__global__ __launch_bounds__(64) void da_split_scores_kernel(const float* __restrict__ Q, int ldq, const cz_seq_desc* __restrict__ descs, int n_heads, int rep,
                                                             size_t kv_off, const int* __restrict__ pos_dev, float* __restrict__ SC /*[S][H][2048]*/,
                                                             float* __restrict__ CM /*[S][H][32]*/) {
    __shared__ __attribute__((aligned(16))) float qs[64];
    cz_seq_desc sd = descs[blockIdx.x];
    const int pos = pos_dev ? *pos_dev : sd.pos0;
    const int nk = pos + 1, c = blockIdx.z, h = blockIdx.y, lane = threadIdx.x;
    if (c * 64 >= nk) return;
    const int kvh = h / rep;
    gfloat* Kb = (gfloat*)(sd.k + kv_off + (sd.kt_stride ? (size_t)kvh * 64 * sd.kt_stride : (size_t)kvh * 64));
    const int j = c * 64 + lane, jc = j < nk ? j : nk - 1;                       // (clamped address, the value is discarded below)
    float kreg[64];
    if (sd.kt_stride) {
#pragma unroll
        for (int d = 0; d < 64; ++d) kreg[d] = __builtin_nontemporal_load(Kb + (size_t)d * sd.kt_stride + jc);
    } else {
        const __attribute__((address_space(1))) f32x4* kr = (const __attribute__((address_space(1))) f32x4*)(Kb + (size_t)jc * sd.kv_stride);
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) { const f32x4 t4 = kr[q4]; kreg[4 * q4] = t4[0]; kreg[4 * q4 + 1] = t4[1]; kreg[4 * q4 + 2] = t4[2]; kreg[4 * q4 + 3] = t4[3]; }
    }
    qs[lane] = Q[(size_t)sd.row0 * ldq + h * 64 + lane];
    __syncthreads();
    float sacc = 0.0f;
#pragma unroll
    for (int q4 = 0; q4 < 16; ++q4) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + q4 * 4);
        sacc = cz_fma(qv[0], kreg[4 * q4], sacc); sacc = cz_fma(qv[1], kreg[4 * q4 + 1], sacc);
        sacc = cz_fma(qv[2], kreg[4 * q4 + 2], sacc); sacc = cz_fma(qv[3], kreg[4 * q4 + 3], sacc);
    }
    const float sc = j < nk ? sacc * 0.125f : -INFINITY;
    const size_t hb = (size_t)blockIdx.x * n_heads + h;
    SC[hb * (DA_SPLIT_MAXC * 64) + c * 64 + lane] = sc;
    float m = sc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float y = __shfl_xor(m, o); m = y > m ? y : m; }
    if (lane == 0) CM[hb * DA_SPLIT_MAXC + c] = m;
}
__global__ __launch_bounds__(64) void da_split_pv_kernel(const cz_seq_desc* __restrict__ descs, int n_heads, int rep, size_t kv_off, const int* __restrict__ pos_dev,
                                                         const float* __restrict__ SC, const float* __restrict__ CM, float* __restrict__ PART /*[S][H][32][64]*/) {
    __shared__ __attribute__((aligned(16))) float E[DA_SPLIT_MAXC][64];
    __shared__ float csum[DA_SPLIT_MAXC];
    cz_seq_desc sd = descs[blockIdx.x];
    const int pos = pos_dev ? *pos_dev : sd.pos0;
    const int nk = pos + 1, nch = (nk + 63) >> 6, c = blockIdx.z, h = blockIdx.y, lane = threadIdx.x;
    if (c >= nch) return;
    const int kvh = h / rep;
    gfloat* Vb = (gfloat*)(sd.v + kv_off + kvh * 64);
    const size_t hb = (size_t)blockIdx.x * n_heads + h;
    // the V slab of this chunk first (lane = d, 64 keys): it is in flight under the softmax statistics below
    const int jend = nk - c * 64 < 64 ? nk - c * 64 : 64;
    gfloat* vp = Vb + (size_t)(c * 64) * sd.kv_stride + lane;
    float vreg[64];
#pragma unroll
    for (int jx = 0; jx < 64; ++jx) { const int jj = jx < jend ? jx : jend - 1; vreg[jx] = __builtin_nontemporal_load(vp + (size_t)jj * sd.kv_stride); }
    // global maximum (exact in any order)
    float m = lane < nch ? CM[hb * DA_SPLIT_MAXC + lane] : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float y = __shfl_xor(m, o); m = y > m ? y : m; }
    const float* sc = SC + hb * (DA_SPLIT_MAXC * 64);
    float scv[DA_SPLIT_MAXC];
#pragma unroll
    for (int cc = 0; cc < DA_SPLIT_MAXC; ++cc) scv[cc] = sc[(cc < nch ? cc : nch - 1) * 64 + lane];     // every chunk's score of this lane in flight at once (a
                                                                                                        // loop on the runtime chunk count is a chain of L2 round trips)
    float mine = 0.0f;
#pragma unroll
    for (int cc = 0; cc < DA_SPLIT_MAXC; ++cc)
        if (cc < nch) {
            const float e = cz_expf(scv[cc] - m);
            E[cc][lane] = e;
            if (cc == c) mine = e;
        }
    __syncthreads();
    if (lane < nch) {                                                            // canonical chunk sum: acc = acc + e_k, k ascending from +0
        float acc = 0.0f;
#pragma unroll 16
        for (int k = 0; k < 64; ++k) acc = acc + E[lane][k];
        csum[lane] = acc;
    }
    __syncthreads();
    float tot = csum[0];
    for (int cc = 1; cc < nch; ++cc) tot = tot + csum[cc];
    __syncthreads();
    E[0][lane] = mine / tot;                                                     // p of the own chunk (E[0] is free: every lane has read what it needs)
    __syncthreads();
    float acc = 0.0f;
#pragma unroll
    for (int jx = 0; jx < 64; ++jx) acc = cz_fma(E[0][jx], jx < jend ? vreg[jx] : 0.0f, acc);
    PART[(hb * DA_SPLIT_MAXC + c) * 64 + lane] = acc;
}
__global__ __launch_bounds__(64) void da_split_sum_kernel(const cz_seq_desc* __restrict__ descs, int n_heads, const int* __restrict__ pos_dev, const float* __restrict__ PART,
                                                          float* __restrict__ O, int ldo, int out_kq) {
    const cz_seq_desc sd = descs[blockIdx.x];
    const int pos = pos_dev ? *pos_dev : sd.pos0;
    const int nch = (pos + 1 + 63) >> 6, h = blockIdx.y, d = threadIdx.x;
    const float* pp = PART + ((size_t)blockIdx.x * n_heads + h) * DA_SPLIT_MAXC * 64 + d;
    float v[DA_SPLIT_MAXC];
#pragma unroll
    for (int c = 0; c < DA_SPLIT_MAXC; ++c) v[c] = pp[(size_t)(c < nch ? c : nch - 1) * 64];          // (all loads in flight; the clamp keeps them in bounds)
    float t = v[0];
#pragma unroll
    for (int c = 1; c < DA_SPLIT_MAXC; ++c) if (c < nch) t = t + v[c];
    const int col = h * 64 + d;
    O[(size_t)sd.row0 * ldo + (out_kq ? kq_index(col) : col)] = t;
}
size_t cz_dec_attn_split_bytes(int nstreams, int n_heads) { return (size_t)nstreams * n_heads * ((size_t)DA_SPLIT_MAXC * 64 + DA_SPLIT_MAXC + (size_t)DA_SPLIT_MAXC * 64) * sizeof(float); }
int g_dec_attn_split_max = 36;      // tuning knob (cz_debug_set "dec_attn_split_max"): groups with nstreams x n_heads <= this take the three-launch form (0 = never).  Measured at
                                    // 767 keys (profiles/r03_steps_split_attention.jsonl): B = 1: 1.379 ms per step against 1.417 for the one-kernel form, B = 4: 1.474,
                                    // B = 8: 1.566 against 1.543 — the three launches pay for a handful of streams only
#define DA_LONG_W 16
#define DA_LONG_C 8                                   // 16 waves x 8 chunks x 64 keys = 8192 positions
static int da_s_stride(int max_nk) { return ((max_nk + 63) / 64) * 64 + 4; }
static size_t da_smem(int max_nk, int rep) {
    const int nch = (max_nk + 63) / 64;
    return (size_t)(rep * 64 + rep * da_s_stride(max_nk) + nch * rep * 64 + DA_MAXW * DA_MAXREP + rep * 32) * sizeof(float);
}
static size_t da_smem_long(int max_nk, int rep) {
    return (size_t)(rep * 64 + rep * da_s_stride(max_nk) + DA_LONG_W * rep * 64 + DA_MAXW * DA_MAXREP + rep * DA_LONG_W * DA_LONG_C) * sizeof(float);
}
template <int rep, int W>
static hipError_t da_prepare(int max_nk) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_kernel<rep, W, 32 / W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(da_smem(max_nk, rep) + 65536));   // (+ room for the dec_attn_lds_pad tuning knob)
}
template <int rep>
static hipError_t da_prepare_long() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_kernel<rep, DA_LONG_W, DA_LONG_C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)da_smem_long(CZ_KERNEL_MAX_CTX, rep));
}
// per device, at session open (the attribute belongs to the current device's copy of the kernel): everyday variants up to `max_nk` <= 2048 keys,
// the long variant up to CZ_KERNEL_MAX_CTX
hipError_t cz_dec_attn_prepare(int max_nk) {
    if (max_nk > 2048) max_nk = 2048;
    hipError_t e = da_prepare<1, 8>(max_nk);
    if (e == hipSuccess) e = da_prepare<2, 8>(max_nk);
    if (e == hipSuccess) e = da_prepare<3, 8>(max_nk);
    if (e == hipSuccess) e = da_prepare<4, 8>(max_nk);
    if (e == hipSuccess) e = da_prepare<1, 4>(max_nk);
    if (e == hipSuccess) e = da_prepare<2, 4>(max_nk);
    if (e == hipSuccess) e = da_prepare<3, 4>(max_nk);
    if (e == hipSuccess) e = da_prepare<4, 4>(max_nk);
    if (e == hipSuccess) e = da_prepare<1, 16>(max_nk);
    if (e == hipSuccess) e = da_prepare<2, 16>(max_nk);
    if (e == hipSuccess) e = da_prepare<3, 16>(max_nk);
    if (e == hipSuccess) e = da_prepare<4, 16>(max_nk);
    if (e == hipSuccess) e = da_prepare_long<1>();
    if (e == hipSuccess) e = da_prepare_long<2>();
    if (e == hipSuccess) e = da_prepare_long<3>();
    if (e == hipSuccess) e = da_prepare_long<4>();
    return e;
}
int g_dec_attn_small_grid = 128;   // tuning knob (cz_debug_set "dec_attn_small_grid"): grids of at most this many blocks run 16 waves per block
template <int W, int MC>
static void da_launch(hipStream_t st, int rep, dim3 grid, size_t smem, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs,
                      int n_heads, int n_kv, int ss, size_t kv_off, const int* pos_dev, int out_kq, int kv_div) {
    const dim3 block(W * 64);
    switch (rep) {
        case 1: hipLaunchKernelGGL((dec_attn_kernel<1, W, MC>), grid, block, smem, st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, kv_div); break;
        case 2: hipLaunchKernelGGL((dec_attn_kernel<2, W, MC>), grid, block, smem, st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, kv_div); break;
        case 3: hipLaunchKernelGGL((dec_attn_kernel<3, W, MC>), grid, block, smem, st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, kv_div); break;
        default: hipLaunchKernelGGL((dec_attn_kernel<4, W, MC>), grid, block, smem, st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, kv_div); break;
    }
}
int g_dec_attn_waves = 0;           // tuning knob (cz_debug_set "dec_attn_waves"): waves per block of the whole-device grids: 8 (each walks up to 4 chunks; two blocks per CU), 4 (up to 8
                                    // chunks; four blocks per CU) or 0 = four where the group shares the device with other lock-step groups (lanes), eight where it runs alone.  Round 4,
                                    // 512 streams per lane (profiles/r04_dec_attn_four_wave_blocks.txt): attention alone 2.42 / 3.47 / 4.38 ms at 511 / 767 / 1 023 keys with eight waves,
                                    // 2.47 / 3.47 / 4.40 with four (1 536 blocks on 1 024 slots are one and a half generations); a full step alone 6.11-6.15 against 6.17-6.21; TWO lanes
                                    // side by side 9.86-9.98 ms against **9.44-9.49**: the smaller blocks leave the other lane's projection blocks their slots sooner
int g_dec_attn_lds_pad = 0;         // tuning knob (cz_debug_set "dec_attn_lds_pad"): extra dynamic LDS bytes per block of the whole-device grids (fewer blocks per CU)
int g_dec_attn_split_heads = 1;     // tuning knob (cz_debug_set "dec_attn_split_heads"): 0 = never one block per query head
// true when cz_launch_dec_attn takes the three-launch split form for this group
bool cz_dec_attn_is_split(int nstreams, int max_nk, int n_heads, int n_kv, const float* split_scratch, int split_streams) {
    const int rep = n_kv > 0 ? n_heads / n_kv : 0;
    return rep >= 1 && rep <= DA_MAXREP && split_scratch && nstreams <= split_streams && max_nk <= DA_SPLIT_MAXC * 64 && nstreams * n_heads <= g_dec_attn_split_max;
}
// the P.V chunk partials of the split form inside split_scratch: [nstreams][n_heads][DA_SPLIT_MAXC][64] (cz_launch_dec_gemv_attsum adds them up)
const float* cz_dec_attn_split_partials(const float* split_scratch, int split_streams, int n_heads, int* maxc) {
    *maxc = DA_SPLIT_MAXC;
    return split_scratch + (size_t)split_streams * n_heads * DA_SPLIT_MAXC * 64 + (size_t)split_streams * n_heads * DA_SPLIT_MAXC;
}
// out_kq = -1 (split form only): the ordered sum of the chunk partials is left to the consumer (the o-projection of kernels_gemv.hip): O is not written
hipError_t cz_launch_dec_attn(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nstreams, int max_nk,
                              int n_heads, int n_kv, size_t kv_off, const int* pos_dev, int out_kq, float* split_scratch, int split_streams, int concurrent_lanes) {
    if (nstreams <= 0) return hipSuccess;
    const int rep = n_heads / n_kv;
    if (rep < 1 || rep > DA_MAXREP || max_nk > CZ_KERNEL_MAX_CTX) return hipErrorInvalidValue;
    if (out_kq < 0 && !cz_dec_attn_is_split(nstreams, max_nk, n_heads, n_kv, split_scratch, split_streams)) return hipErrorInvalidValue;
    if (split_scratch && nstreams <= split_streams && max_nk <= DA_SPLIT_MAXC * 64 && nstreams * n_heads <= g_dec_attn_split_max) {
        float* SC = split_scratch;
        float* CM = SC + (size_t)split_streams * n_heads * DA_SPLIT_MAXC * 64;
        float* PART = CM + (size_t)split_streams * n_heads * DA_SPLIT_MAXC;
        const int nchm = (max_nk + 63) / 64;
        hipLaunchKernelGGL(da_split_scores_kernel, dim3(nstreams, n_heads, nchm), dim3(64), 0, st, Q, ldq, descs, n_heads, rep, kv_off, pos_dev, SC, CM);
        hipLaunchKernelGGL(da_split_pv_kernel, dim3(nstreams, n_heads, nchm), dim3(64), 0, st, descs, n_heads, rep, kv_off, pos_dev, SC, CM, PART);
        if (out_kq >= 0) hipLaunchKernelGGL(da_split_sum_kernel, dim3(nstreams, n_heads), dim3(64), 0, st, descs, n_heads, pos_dev, PART, O, ldo, out_kq);
        return hipGetLastError();
    }
    const int ss = da_s_stride(max_nk);
    if (max_nk > 2048) {            // contexts past 2048 keys (a re-prime interval > 1536): the 16-wave variant whose waves walk 8 chunks each
        da_launch<DA_LONG_W, DA_LONG_C>(st, rep, dim3(nstreams, n_kv), da_smem_long(max_nk, rep), Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, 1);
        return hipGetLastError();
    }
    if (g_dec_attn_split_heads && rep > 1 && nstreams * n_heads <= g_dec_attn_small_grid) {     // a handful of streams: one block per QUERY head (16 waves): -2 % step time at B = 1
        da_launch<16, 2>(st, 1, dim3(nstreams, n_heads), da_smem(max_nk, 1), Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, rep);
        return hipGetLastError();
    }
    const dim3 grid(nstreams, n_kv);
    const size_t smem = da_smem(max_nk, rep);
    if (nstreams * n_kv <= g_dec_attn_small_grid) da_launch<16, 2>(st, rep, grid, smem, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, 1);
    else if (g_dec_attn_waves == 4 || (g_dec_attn_waves == 0 && concurrent_lanes > 1)) da_launch<4, 8>(st, rep, grid, smem + (size_t)g_dec_attn_lds_pad, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, 1);
    else da_launch<8, 4>(st, rep, grid, smem + (size_t)g_dec_attn_lds_pad, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev, out_kq, 1);
    return hipGetLastError();
}

// kernels_gemm_mid.hip — C[M,N] = A[M,K] (f32) x W[N,K]^T (bf16) for a FEW HUNDRED rows: the projections of one wide lock-step decode
// group (M = 256..1024 streams), in the canonical reduction order (chunks of 64 k, fmaf chain inside a chunk == an MFMA chain, chunk
// partials added left to right in f32).
//
// Why a third GEMM shape.  At these row counts a projection is 0.3-2.7 GFLOP: 2-17 us of the f32 MFMA peak, spread over 256 CUs only if
// it is cut into ~500+ pieces.  The big-M kernel's tiles (>= 64 x 64, K walked serially) give 54-288 blocks that each walk 9-24 K tiles
// alone on a CU; dec_gemm_kernel splits K over the waves of a block but re-reads the activations once per 32 output columns and keeps
// one accumulator chain per operand load.  Here:
//   * block = ONE 32 x (32 TN) output tile, NW waves = NW K GROUPS: wave g owns K tiles g, g+NW, ...  (NW = 4 by default; see launch_mid).
//     The M/32 x N/(32 TN) grid is 432 blocks for the 768 x 576 o-projection, 1152 for gate|up.
//   * the ordered sum of the chunk partials crosses the waves through LDS: every round (NW K tiles) waves 1.. park their partial tiles
//     on top of their own (spent) activation tile, ONE barrier, and wave 0 adds them — in K order — to the running total; where there
//     is a next round it does so at its start and then raises an LDS flag, which the other waves await before their next activation
//     tile replaces the parked partial.  35 KB of LDS per four-wave block: four blocks share a CU (two for gate|up: 180 registers).
//   * weights are read straight into MFMA operand registers from a copy packed for v_mfma_f32_32x32x2_f32 ("k2": per row and 64-k chunk
//     the even k, then the odd k, so that lane (n, h) loads its 32 operands k = 2t + h as 64 contiguous bytes); only the activations
//     pass through LDS (a wave-private 32 x 64 tile: no block barrier for staging).
// Bound: f32 MFMA (157.3 TFLOP/s).  Algorithmic flops per launch = 2 M N K.
#include "cz_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tuning build (make mid_dbg, tools/mid_phases.py): wave 0 of the FIRST and of the LAST block of a launch accumulate s_memrealtime ticks (10 ns) per
// phase; every stamp first waits for the wave's outstanding memory operations
#ifdef CZ_MID_STAMPS
__device__ unsigned long long g_mid_stamp[8][16];
#define MST_BEGIN const int mst_k = (EPI & 3) + ((int)blockIdx.x == (int)gridDim.x - 1 ? 4 : ((int)blockIdx.x == 0 ? 0 : -100)); unsigned long long mst_t = 0; \
                  if (mst_k >= 0 && threadIdx.x == 0) mst_t = wall_clock64()
#define MST(ph) do { if (mst_k >= 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = wall_clock64(); atomicAdd(&g_mid_stamp[mst_k][ph], t_ - mst_t); \
                     if ((ph) == 15) atomicAdd(&g_mid_stamp[mst_k][0], 1ull); mst_t = t_; } } while (0)
extern "C" void cz_debug_mid_stamps(unsigned long long* out128, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_mid_stamp), sizeof(unsigned long long) * 128);
    if (reset) { unsigned long long z[128] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_mid_stamp), z, sizeof z); }
}
#else
#define MST_BEGIN do { } while (0)
#define MST(ph) do { } while (0)
#endif

#define MID_LDK 68   // LDS row stride of the activation tile in floats (64 + 4: ds_read_b128 by 16 consecutive rows is conflict-free)

// Wk[n'][c*64 + h*32 + t] = W[n][c*64 + 2t + h].  The first `rot_heads` 64-row heads are stored with their rows reordered so that a
// 32-column output block holds 16 RoPE pairs (j, j + 32) of ONE head: n' = 64 head + 32 b + r takes row 16 b + (r & 15) + 32 (r >> 4).
__global__ void k2_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wk, size_t N, int K, int rot_heads) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N * (size_t)K; i += (size_t)gridDim.x * blockDim.x) {
        const size_t np = i / K; const int k = (int)(i % K);
        size_t n = np;
        if (np < (size_t)rot_heads * 64) { const int j = (int)(np & 63), b = j >> 5, r = j & 31; n = (np & ~(size_t)63) + 16 * b + (r & 15) + 32 * (r >> 4); }
        Wk[np * K + (k & ~63) + (k & 1) * 32 + ((k & 63) >> 1)] = W[n * K + k];
    }
}
hipError_t cz_launch_k2_pack(hipStream_t st, const uint16_t* W, uint16_t* Wk, size_t N, int K, int rot_heads) {
    if (K % 64 || (size_t)rot_heads * 64 > N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k2_pack_kernel, dim3(2048), dim3(256), 0, st, W, Wk, N, K, rot_heads);
    return hipGetLastError();
}
struct mid_rope_args {          // EPI = 2: the fused-QKV projection of one lock-step decode step
    const cz_rope_desc* descs;  // per stream (= output row): where its K / V cache lives
    const float* cs; const float* sn;
    const int* pos_dev;
    size_t kv_off;
    int n_heads, n_kv;
};

__device__ __forceinline__ int mid_xcd_remap(int bid, int nblocks) {    // blocks b and b + 8 share an XCD (speed only): contiguous tile ranges per XCD
    int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
    int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// EPI = 0: C = A.W^T (+ R).  EPI = 1 (TN = 2): W rows interleaved gate|up by 32 (cz_launch_gu_interleave): H[M][N/2] = silu(gate) * up.
// EPI = 2 (TN = 1): the fused QKV projection with the head rows reordered by cz_launch_k2_pack(rot_heads = n_heads + n_kv): RoPE
// (candle_nn::rotary_emb::rope, non-interleaved halves, products rounded separately) in the epilogue; q goes to C, k (rotated) and v
// straight to the stream's KV cache at *pos_dev — what rope_kernel did in a launch of its own.
// EPI = 3: C = relu(A.W^T)^2 (the RWKV7 channel-mix key projection, rwkv7.rs:426).
// The tile (bm, bn) of one problem; the kernels below pick the tile.
// QE ("quad epilogue"): the ordered sum of the chunk partials AND the epilogue are shared by waves 0..3 — wave w owns registers 4w..4w+3 of the
// 32 x 32 C/D tile (rows 8w + {0..3} + 4 (lane >> 5)): every wave parks its partial tile, one barrier, wave w adds ITS quad of every parked tile in
// K order (the same left-to-right order: an element's sum does not care which wave forms it), a second barrier frees the tiles.  With !QE wave 0
// alone adds all sixteen registers of every partial and runs the whole epilogue (RoPE + cache scatter, SwiGLU's exp and division: 3-4 us of a
// 14-22 us block, profiles/r03_mid_phases_b512_before_quad_epilogue.json) while waves 1..3 have nothing to do.
template <int TN, int EPI, int NW, bool PF, bool QE>
__device__ __forceinline__ void gemm_mid_body(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Wk,
                                              float* __restrict__ C, int ldc, const float* __restrict__ R, int ldr,
                                              int M, int N, int K, const int bm, const int bn, const mid_rope_args& ra_) {
    static_assert(TN * 1024 <= 32 * MID_LDK, "a wave's partial tiles are parked in its own activation tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* As = smem + wave * (32 * MID_LDK);                            // wave-private activation tile [32][MID_LDK]; after the chain: this wave's partial tiles
    int* consumed = reinterpret_cast<int*>(smem + NW * (32 * MID_LDK));  // round up to which wave 0 has taken the partials over
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = K >> 6, rounds = (nk + NW - 1) / NW;
    const int srow = lane >> 3, sg = lane & 7;                           // staging: 8 rows x 8 k-groups per pass, 4 passes

    f32x4 ra[4][2];
    uint4 rb[TN][4];
    // Addresses as in gemm_bf16w_kernel (round 4): a buffer descriptor at the tile's origin + a per-lane byte offset computed once + a scalar offset
    // per K tile — the loads of a round cost no vector instruction besides themselves (a vector instruction beside the other waves' MFMA streams
    // costs ~35 cycles, and twelve waves share a CU here).  Rows past M read row M - 1 (an output row depends on its own activation row only and
    // the epilogues store rows < M only).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)(bm * 32) * lda), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Wk + (size_t)(bn * (32 * TN)) * K), 0, 0xffffffffu, 0x00020000);
    const int mrem = M - 1 - bm * 32;
    unsigned aoff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) aoff[p] = ((unsigned)min(p * 8 + srow, mrem) * (unsigned)lda + sg * 8) * 4u;
    const unsigned woff = ((unsigned)lr * (unsigned)K + lh * 32) * 2u;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[p], kt * 256, 0));
            ra[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[p], kt * 256 + 16, 0));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto w = __builtin_amdgcn_raw_buffer_load_b128(rsW, woff, (j * 32 * K + kt * 64) * 2 + q * 16, 0);
                rb[j][q] = uint4{w[0], w[1], w[2], w[3]};
            }
    };
    // LDS image (as gemm_bf16w_kernel): within every group of 8 k the even k first, then the odd k, so that the b128 read of lane-half h
    // at 8g + 4h yields k = 8g + 2j + h, j = 0..3 = the A operands of MFMA steps 4g + j
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* dst = As + (p * 8 + srow) * MID_LDK + sg * 8;
            *reinterpret_cast<f32x4*>(dst) = f32x4{ra[p][0][0], ra[p][0][2], ra[p][1][0], ra[p][1][2]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{ra[p][0][1], ra[p][0][3], ra[p][1][1], ra[p][1][3]};
        }
    };
    // a wave's partial tiles in LDS: [TN][4][64 lanes][4] floats (b128 per lane and register quad)
    auto add_partials = [&](f32x16* tot, int first_chunk) {             // wave 0: + the partials of waves 1..NW-1 (chunks first_chunk + w), in K order
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (first_chunk + w < nk) {
                const float* sp = smem + w * (32 * MID_LDK);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + ((j * 4 + q) * 64 + lane) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[j][4 * q + e] = tot[j][4 * q + e] + v[e];
                    }
            }
    };

    f32x16 tot[TN];
    if (PF || QE) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[j][r] = 0.0f;
    }
    MST_BEGIN;
    if (wave < nk) load_tile(wave);
    MST(1);                                                              // first operand tile arrived
    for (int rd = 0; rd < rounds; ++rd) {
        const int kt = rd * NW + wave;
        if (PF && !QE) {
            if (wave == 0) {
                if (rd > 0) add_partials(tot, (rd - 1) * NW);             // (complete since the barrier that ended the previous round)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(consumed, rd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (rd > 0 && kt < nk) {                              // this wave's tile is about to replace its parked partial: wave 0 must have it
                while (__hip_atomic_load(consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rd) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        MST(2);                                                          // (wave 0) partials of the previous round added, flag raised
        if (kt < nk) {
            store_tile();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);                        // this wave's LDS writes are visible to its own lanes (lgkmcnt(0))
            unsigned wb[TN][16];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) { wb[j][4 * q] = rb[j][q].x; wb[j][4 * q + 1] = rb[j][q].y; wb[j][4 * q + 2] = rb[j][q].z; wb[j][4 * q + 3] = rb[j][q].w; }
            if (PF && kt + NW < nk) load_tile(kt + NW);                 // next tile of this wave in flight under the MFMA chain
            f32x16 acc[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(As + lr * MID_LDK + g * 8 + lh * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int t = 4 * g + s;                              // MFMA step: k = 2t + lh
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned w = wb[j][t >> 1];
                        const float bv = (t & 1) ? cz_bits_f32(w & 0xffff0000u) : cz_bits_f32(w << 16);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bv, acc[j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);                        // (the reads of As are done before the partial goes on top of it)
            if (!QE && wave == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[j][r] = (!PF || rd == 0) ? acc[j][r] : tot[j][r] + acc[j][r];
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(As + ((j * 4 + q) * 64 + lane) * 4) = f32x4{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
            }
        }
        MST(3);                                                          // staging + MFMA chain of the round (the next tile's loads waited for too)
        __syncthreads();                                                // one barrier per round: the partials of round rd are complete
        MST(4);                                                          // barrier
        if (QE) {
            if (wave < 4) {                                               // quad `wave` of every tile parked this round, chunks rd * NW + v in order
#pragma unroll
                for (int v = 0; v < NW; ++v)
                    if (rd * NW + v < nk) {
                        const float* sp = smem + v * (32 * MID_LDK);
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(sp + ((j * 4 + wave) * 64 + lane) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) tot[j][e] = (rd == 0 && v == 0) ? q[e] : tot[j][e] + q[e];      // (tot[j][0..3] = this wave's quad)
                        }
                    }
            }
            MST(5);
            if (rd + 1 < rounds) __syncthreads();                        // the parked tiles are free for the next round's activations
        }
    }
    if (QE) { if (wave >= 4) return; }
    else {
        if (wave != 0) return;
        add_partials(tot, (rounds - 1) * NW);
        MST(5);                                                              // last round's partials added
    }
    // QE: this wave holds registers 4 wave .. 4 wave + 3 of the tile in tot[j][0..3]; !QE: wave 0 holds all sixteen
    constexpr int NR = QE ? 4 : 16;
    const int r0 = QE ? 4 * wave : 0;
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (EPI == 2) {
        const int pos = ra_.pos_dev ? *ra_.pos_dev : ra_.descs[0].pos0;
        const int nrot = (ra_.n_heads + ra_.n_kv) * 64, col0 = bn * 32;
        if (col0 < nrot) {
            const int h = bn >> 1, t = (bn & 1) * 16 + (lr & 15);
            const bool lo = lr < 16;                                     // lanes lr and lr ^ 16 hold the pair (t, t + 32) of head h
            const float c = ra_.cs[(size_t)pos * 32 + t], sn = ra_.sn[(size_t)pos * 32 + t];
            const int dim = t + (lo ? 0 : 32);
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int r = r0 + e;
                const int row = bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float mine = tot[0][e], other = __shfl_xor(mine, 16);
                const float x1 = lo ? mine : other, x2 = lo ? other : mine;
                const float pa = x1 * (lo ? c : sn), pb = x2 * (lo ? sn : c);      // lo: x1 c, x2 s; hi: x1 s, x2 c
                const float o = lo ? pa - pb : pa + pb;
                if (row < M) {
                    if (h < ra_.n_heads) C[(size_t)row * ldc + h * 64 + dim] = o;
                    else {
                        const cz_rope_desc d = ra_.descs[row];
                        const int kd = (h - ra_.n_heads) * 64 + dim;
                        if (d.kt_stride) d.k_dst[ra_.kv_off + (size_t)kd * d.kt_stride + pos] = o;
                        else d.k_dst[ra_.kv_off + (size_t)pos * d.dst_stride + kd] = o;
                    }
                }
            }
        } else {
            const int vc = col0 - nrot + lr;
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int r = r0 + e;
                const int row = bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) { const cz_rope_desc d = ra_.descs[row]; d.v_dst[ra_.kv_off + (size_t)pos * d.dst_stride + vc] = tot[0][e]; }
            }
        }
        MST(15);
        return;
    }
    const bool full = bm * 32 + 32 <= M;                                 // (block-uniform) every row of the tile is stored: scalar row offsets, no predicate
    if (EPI == 1) {
        if (full) {
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)(bm * 32) * ldc + (bn * (32 * TN)) / 2, 0, 0xffffffffu, 0x00020000);
            const unsigned loff = ((unsigned)(4 * lh) * (unsigned)ldc + lr) * 4u;
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int r = r0 + e;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cz_siluf(tot[0][e]) * tot[TN - 1][e]), rsC, loff, ((r & 3) + 8 * (r >> 2)) * ldc * 4, 0);
            }
            MST(15);
            return;
        }
        const int hcol = (bn * (32 * TN)) / 2 + lr;
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            const int r = r0 + e;
            const int row = bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < M) C[(size_t)row * ldc + hcol] = cz_siluf(tot[0][e]) * tot[TN - 1][e];      // silu(gate) * up (llama.rs MLP)
        }
        MST(15);
        return;
    }
    if (full) {
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)(bm * 32) * ldc + bn * (32 * TN), 0, 0xffffffffu, 0x00020000);
        const unsigned loff = ((unsigned)(4 * lh) * (unsigned)ldc + lr) * 4u;
        if (R) {
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + (size_t)(bm * 32) * ldr + bn * (32 * TN)), 0, 0xffffffffu, 0x00020000);
            const unsigned roff = ((unsigned)(4 * lh) * (unsigned)ldr + lr) * 4u;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float rv[NR];
#pragma unroll
                for (int e = 0; e < NR; ++e) { const int r = r0 + e; rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, roff, (((r & 3) + 8 * (r >> 2)) * ldr + j * 32) * 4, 0)); }
#pragma unroll
                for (int e = 0; e < NR; ++e) { const int r = r0 + e; __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[j][e] + rv[e]), rsC, loff, (((r & 3) + 8 * (r >> 2)) * ldc + j * 32) * 4, 0); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const int r = r0 + e;
                    float v = tot[j][e];
                    if (EPI == 3) { const float z = v > 0.0f ? v : 0.0f; v = z * z; }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, loff, (((r & 3) + 8 * (r >> 2)) * ldc + j * 32) * 4, 0);
                }
        }
        MST(15);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = bn * (32 * TN) + j * 32 + lr;
        float rv[NR];                                                     // (all residual loads in flight together: see gemm_bf16w_kernel)
        if (R) {
#pragma unroll
            for (int e = 0; e < NR; ++e) { const int r = r0 + e; rv[e] = R[(size_t)min(bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1) * ldr + col]; }
        }
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            const int r = r0 + e;
            const int row = bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = R ? tot[j][e] + rv[e] : tot[j][e];
            if (EPI == 3) { const float z = tot[j][e] > 0.0f ? tot[j][e] : 0.0f; v = z * z; }
            if (row < M) C[(size_t)row * ldc + col] = v;
        }
    }
    MST(15);
}

template <int TN, int EPI, int NW, bool PF, bool QE>
__global__ __launch_bounds__(NW * 64, (TN == 1 && NW == 9) ? 5 : ((TN == 2 && NW == 4 && QE) ? 3 : 1)) void gemm_mid_kernel(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Wk,
                                                           float* __restrict__ C, int ldc, const float* __restrict__ R, int ldr,
                                                           int M, int N, int K, int nbx, int nblocks, int colmaj, mid_rope_args ra_) {
    // An XCD (blocks b, b + 8, ...: speed only) takes a contiguous range of tiles.  Row-major numbering gives it a few row tiles x ALL column
    // tiles: its L2 fetches the whole weight panel and 1/8 of the activations; column-major (colmaj) a few column tiles x all row tiles: 1/8 of
    // the weights, all the activations.  The launcher picks the order that fetches fewer bytes over the eight L2s (gate|up at 512 rows:
    // 3.5 MB of weights x 8 against 1.2 MB of activations x 8).
    const int sid = mid_xcd_remap(blockIdx.x, nblocks);
    const int nby = nblocks / nbx;
    const int bm = colmaj ? sid % nby : sid / nbx, bn = colmaj ? sid / nby : sid % nbx;
    gemm_mid_body<TN, EPI, NW, PF, QE>(A, lda, Wk, C, ldc, R, ldr, M, N, K, bm, bn, ra_);
}

// Several independent plain GEMMs that share M in ONE launch (the RWKV7 time-mix: seven projections of differently mixed inputs, then four
// LoRA up-projections): tile t of the launch belongs to the problem z with tile0[z] <= t < tile0[z + 1]; column-major inside a problem (a
// few hundred rows against 768-wide weight panels: the weights are the larger operand).
#define MID_GROUP_MAX 8
struct mid_prob { const float* A; const uint16_t* Wk; float* C; int lda, ldc, N, K; };
struct mid_group { mid_prob p[MID_GROUP_MAX]; int tile0[MID_GROUP_MAX + 1]; int n; };
template <int NW, bool PF>
__global__ __launch_bounds__(NW * 64, 1) void gemm_mid_group_kernel(mid_group g, int M, int nblocks) {
    const int sid = mid_xcd_remap(blockIdx.x, nblocks);
    int z = 0;
#pragma unroll
    for (int q = 1; q < MID_GROUP_MAX; ++q) if (q < g.n && sid >= g.tile0[q]) z = q;
    const mid_prob pr = g.p[z];
    const int local = sid - g.tile0[z], nby = (M + 31) / 32;
    gemm_mid_body<1, 0, NW, PF, true>(pr.A, pr.lda, pr.Wk, pr.C, pr.ldc, nullptr, 0, M, pr.N, pr.K, local % nby, local / nby, mid_rope_args{});
}

int g_mid_quad = 1;           // tuning knob (cz_debug_set "mid_quad"): 1 = the four-wave blocks share the ordered sum and the epilogue (QE), 0 = wave 0 does both
int g_mid_colmaj = -1;        // tuning knob (cz_debug_set "mid_colmaj"): -1 = by operand sizes, 0 = row-major tile order, 1 = column-major
template <int TN, int EPI, int NW, bool PF>
static hipError_t launch_mid_nw(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, const float* R, int ldr, int M, int N, int K,
                                mid_rope_args ra) {
    const size_t smem = (size_t)(NW * 32 * MID_LDK + 4) * sizeof(float);
    auto kern = g_mid_quad ? gemm_mid_kernel<TN, EPI, NW, PF, true> : gemm_mid_kernel<TN, EPI, NW, PF, false>;   // (LDS limit: cz_gemm_mid_prepare, once per device at session open)
    const int nbx = N / (32 * TN), nby = (M + 31) / 32, nblocks = nbx * nby;
    // bytes fetched over the eight XCD L2s: row-major = 8 x weights + activations, column-major = weights + 8 x activations
    const double wb = (double)N * K * 2, ab = (double)M * K * 4;
    const int colmaj = g_mid_colmaj < 0 ? (wb + 8 * ab < 8 * wb + ab ? 1 : 0) : g_mid_colmaj;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64), smem, st, A, lda, Wk, C, ldc, R, ldr, M, N, K, nbx, nblocks, colmaj, ra);
    return hipGetLastError();
}
template <int TN, int EPI, int NW, bool PF>
static hipError_t mid_prepare_one() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mid_kernel<TN, EPI, NW, PF, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((size_t)(NW * 32 * MID_LDK + 4) * sizeof(float)));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mid_kernel<TN, EPI, NW, PF, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)((size_t)(NW * 32 * MID_LDK + 4) * sizeof(float)));
    return e;
}
template <int TN, int EPI>
static hipError_t mid_prepare_epi() {
    hipError_t e = mid_prepare_one<TN, EPI, 4, true>();
    if (e == hipSuccess) e = mid_prepare_one<TN, EPI, 8, true>();
    if (e == hipSuccess) e = mid_prepare_one<TN, EPI, 9, false>();
    if (e == hipSuccess) e = mid_prepare_one<TN, EPI, TN == 1 ? 12 : 8, true>();
    return e;
}
hipError_t cz_gemm_mid_prepare() {
    hipError_t e = mid_prepare_epi<1, 0>();
    if (e == hipSuccess) e = mid_prepare_epi<2, 1>();
    if (e == hipSuccess) e = mid_prepare_epi<1, 2>();
    if (e == hipSuccess) e = mid_prepare_epi<1, 3>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mid_group_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)((size_t)(4 * 32 * MID_LDK + 4) * sizeof(float)));
    return e;
}
// K groups per block.  Round 3, with the ordered sum and the epilogue shared by four waves (QE), 512 rows, ms per decode step
// (profiles/r03_steps_mid_k_groups.jsonl): four groups 6.29-6.36, five (K = 576 in two rounds instead of 4 + 4 + 1) 6.39, six 6.47, eight 6.60, one
// wave per chunk (nine / twelve) 6.45; wave 0 doing sum and epilogue alone (!QE) 6.62.  Round 2, measured on 512-row groups (profiles/r02_microbench/call24_*): four groups with operand prefetch beat a wave per
// 64-chunk (nine waves, one round for K = 576: +8 % step time) and summing on the round's last wave instead of its first (the wave with
// the fewest K tiles: +2 %, call26) — a block's latency is dominated by its first, cold operand load and by
// the 0.85 us dependent MFMA chain per chunk, and four-wave blocks pack four to a CU where nine-wave blocks pack one or two.
int g_mid_nw = 4;             // tuning knob (cz_debug_set "mid_nw"): 4 / 8 = that many K groups with operand prefetch, 0 = nine (K <= 576, one round) or twelve
template <int TN, int EPI>
static hipError_t launch_mid(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, const float* R, int ldr, int M, int N, int K,
                             mid_rope_args ra = mid_rope_args{}) {
    if (g_mid_nw == 8) return launch_mid_nw<TN, EPI, 8, true>(st, A, lda, Wk, C, ldc, R, ldr, M, N, K, ra);
    if (g_mid_nw != 0) return launch_mid_nw<TN, EPI, 4, true>(st, A, lda, Wk, C, ldc, R, ldr, M, N, K, ra);
    if (K / 64 <= 9) return launch_mid_nw<TN, EPI, 9, false>(st, A, lda, Wk, C, ldc, R, ldr, M, N, K, ra);
    return launch_mid_nw<TN, EPI, TN == 1 ? 12 : 8, true>(st, A, lda, Wk, C, ldc, R, ldr, M, N, K, ra);   // (two accumulator tiles: 168 registers do not hold a 12-wave block's third wave per SIMD)
}
// C[M][N] = A . W^T (+ R); Wk = the k2-packed copy of W
hipError_t cz_launch_gemm_mid(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, const float* R, int ldr, int M, int N, int K) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || N % 32 != 0) return hipErrorInvalidValue;
    return launch_mid<1, 0>(st, A, lda, Wk, C, ldc, R, ldr, M, N, K);
}
// H[M][f] = silu(A . Wgate^T) * (A . Wup^T); Wk = the k2-packed copy of the gate|up rows interleaved by 32 (cz_launch_gu_interleave)
hipError_t cz_launch_gemm_mid_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* H, int ldh, int M, int f, int K) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || f % 32 != 0) return hipErrorInvalidValue;
    return launch_mid<2, 1>(st, A, lda, Wk, H, ldh, nullptr, 0, M, 2 * f, K);
}
// The fused QKV projection of a lock-step decode step: Q[M][n_heads*64] (rotated) to C, K (rotated) and V to the KV cache rows at *pos_dev.
// Wk = cz_launch_k2_pack(..., rot_heads = n_heads + n_kv) of the [q | k | v] weight rows.
hipError_t cz_launch_gemm_mid_qkv(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* Q, int ldq, int M, int K, int n_heads, int n_kv,
                                  const cz_rope_desc* descs, size_t kv_off, const int* pos_dev, const float* cs, const float* sn) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || !descs || !cs || !sn) return hipErrorInvalidValue;
    mid_rope_args ra{descs, cs, sn, pos_dev, kv_off, n_heads, n_kv};
    return launch_mid<1, 2>(st, A, lda, Wk, Q, ldq, nullptr, 0, M, (n_heads + 2 * n_kv) * 64, K, ra);
}

// C[M][N] = relu(A . W^T)^2
hipError_t cz_launch_gemm_mid_relusq(hipStream_t st, const float* A, int lda, const uint16_t* Wk, float* C, int ldc, int M, int N, int K) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || N % 32 != 0) return hipErrorInvalidValue;
    return launch_mid<1, 3>(st, A, lda, Wk, C, ldc, nullptr, 0, M, N, K);
}
// nprob plain GEMMs C_i[M][N_i] = A_i . W_i^T sharing M, one launch (four K groups per block)
hipError_t cz_launch_gemm_mid_group(hipStream_t st, int nprob, const float* const* A, const int* lda, const uint16_t* const* Wk, float* const* C, const int* ldc,
                                    const int* N, const int* K, int M) {
    if (M <= 0 || nprob <= 0) return hipSuccess;
    if (nprob > MID_GROUP_MAX) return hipErrorInvalidValue;
    mid_group g{};
    const int nby = (M + 31) / 32;
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        if (N[i] % 32 || K[i] % 64) return hipErrorInvalidValue;
        g.p[i] = mid_prob{A[i], Wk[i], C[i], lda[i], ldc[i], N[i], K[i]};
        g.tile0[i] = tiles; tiles += nby * (N[i] / 32);
    }
    for (int i = nprob; i <= MID_GROUP_MAX; ++i) g.tile0[i] = tiles;
    g.n = nprob;
    const size_t smem = (size_t)(4 * 32 * MID_LDK + 4) * sizeof(float);
    hipLaunchKernelGGL((gemm_mid_group_kernel<4, true>), dim3(tiles), dim3(4 * 64), smem, st, g, M, tiles);
    return hipGetLastError();
}

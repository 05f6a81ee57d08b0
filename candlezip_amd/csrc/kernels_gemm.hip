// kernels_gemm.hip — C[M,N] = A[M,K] (f32) x W[N,K]^T (bf16, widened exactly) in the canonical reduction order.
//
// Replaces the Linear layers candle runs through the `gemm` crate on the reference CPU path (call sites
// src/models.rs:94, :110; candle_rwkv7/src/models/rwkv7.rs:206-236, 324, 426-428, 518).
//
// Canonical order (DESIGN.md "Numerics"): K is cut into chunks of 64; inside a chunk the partial sum is the
// fmaf chain k ascending from +0 — which is bit for bit what a chain of v_mfma_f32_32x32x2_f32 / 16x16x4_f32
// computes — and chunk partials are added left to right in f32 on the VALU.  One LDS K-tile == one chunk.
//
// gemm_bf16w_kernel   : big-M (prefill / segment-parallel encode).  Bound: f32 MFMA (157.3 TFLOP/s peak).
//                       Algorithmic flops per launch = 2*M*N*K.
// gemm_skinny_kernel  : M <= 64 (lock-step decode).  K chunks are spread over the waves of a block, the chunk
//                       partials meet in LDS and are summed in chunk order.  Latency/HBM-bound.
#include "cz_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDK 68   // LDS row stride in floats (64 + 4): ds_read_b128 by 16 consecutive rows is conflict-free

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    // blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness):
    // give each XCD a contiguous range of tiles so that tiles sharing an A row-panel hit one L2.
    int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
    int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// EPI = 1 (gate|up projection of the MLP): W holds the gate and up rows interleaved in groups of 32 (cz_launch_gu_interleave), so the two
// 32-column MFMA tiles of a wave are the gate and the up values of the SAME 32 hidden columns and the epilogue writes
// silu(gate) * up straight into H [M][N/2] — the 2 x f wide intermediate never exists in HBM.
// tuning build (make gemm_clock, tools/gemm_clock.py): thread 0 of every 64th block accumulates shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) over its
// K loop: their ratio is the clock the chip holds under this kernel's load (MI355X_MICROARCH "DVFS give-back" item 6) — the f32 MFMA peak AT THAT CLOCK is what the
// kernel can be priced against.  The stamps go to a buffer of their own; no output depends on them.
#ifdef CZ_GEMM_CLOCK
__device__ unsigned long long g_gemm_clock[32];      // [0] shader ticks in K loops, [1] 100 MHz ticks in K loops, [2] sampled blocks, [3] their K tiles, [4] shader ticks entry -> K loop (first tile staged), [5] K loop end -> stores issued; [8..13] the same for the BN = 64 tiles
extern "C" void cz_debug_gemm_clock(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gemm_clock), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_clock), z, sizeof z); }
}
#endif
#define LDH 32   // LDS row stride of a HALF K tile, PIPE form: unpadded; the 16-byte column c of row r lives at column c ^ ((r >> 1) & 7) — conflict-free
                 // ds_read_b128 by 16 consecutive rows and ds_write_b128 by 4 rows x 4 column pairs, and three 128 x 64 blocks (3 x 48 KB) share a CU
// PIPE (round 4): the K tile is staged as two half tiles of 32 k in two LDS buffers, so that the staging of half h + 1 (bf16 widening, even/odd
// deinterleave, ds_write) is issued BETWEEN this wave's own MFMAs on half h instead of alone between two barriers: a vector instruction next to
// the wave's own MFMAs costs the matrix pipe ~3 cycles, alone beside the partner block's MFMA stream ~35 (profiles/r04_gemm_phases/: staging was
// 3.2 k of the 17.1 k cycles of a K tile).  One barrier per half tile; the accumulator chain of a 64-chunk runs through both halves, so the
// arithmetic is unchanged.
template <int BM, int BN, int WM, int WN, int EPI, bool PIPE = false>
__global__ __launch_bounds__(256, (PIPE && BM + BN <= 192) ? 3 : 2) void gemm_bf16w_kernel(const float* __restrict__ A, int lda,
                                                         const uint16_t* __restrict__ W, float* __restrict__ C, int ldc,
                                                         const float* __restrict__ R, int ldr, int M, int N, int K,
                                                         int nbx, int nblocks, const unsigned char* __restrict__ row_mask) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = BM / 32, PB = BN / 32;     // staging passes (256 threads x 8 k-values per pass)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + BM * LDK;

    const int sid = xcd_remap(blockIdx.x, nblocks);
    const int bm = sid / nbx, bn = sid % nbx;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the wave index in an SGPR: everything derived from it stays scalar)
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    f32x4 ra[PA][2];
    uint4 rb[PB];
    const int srow = tid >> 3, sg = tid & 7;      // staging: row within a 32-row pass, 8-wide k group

    // Addresses as buffer descriptor (block-uniform tile origin, SGPRs) + per-lane 32-bit byte offset computed ONCE + scalar offset: the loads of the K loop and
    // the stores of the epilogue then cost no vector instruction besides themselves (`buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`).  A co-resident
    // block's MFMA stream leaves this wave about two vector issues per 64-cycle MFMA (round 4, profiles/r04_gemm_phases/): every address VALU in the prologue,
    // the loop and the epilogue is paid ~30 cycles apiece, and the 64-bit address arithmetic of plain pointers was most of the epilogue's 47 k cycles.
    // Rows past M re-read row M - 1: their results are never stored.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)(bm * BM) * lda), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W + (size_t)(bn * BN) * K), 0, 0xffffffffu, 0x00020000);
    const int mrem = M - 1 - bm * BM;             // last valid row of this tile (>= 0)
    unsigned aoff[PA];                            // byte offsets inside the tile's rows
#pragma unroll
    for (int p = 0; p < PA; ++p) aoff[p] = ((unsigned)min(p * 32 + srow, mrem) * (unsigned)lda + sg * 8) * 4u;
    const unsigned woff = ((unsigned)srow * (unsigned)K + sg * 8) * 2u;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            ra[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[p], kt * 256, 0));
            ra[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[p], kt * 256 + 16, 0));
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const auto w = __builtin_amdgcn_raw_buffer_load_b128(rsW, woff, (p * 32 * K + kt * 64) * 2, 0);
            rb[p] = uint4{w[0], w[1], w[2], w[3]};
        }
    };
    // LDS image: within every group of 8 k-values the even k come first, then the odd k, so that the b128 read of
    // lane-half h (4 floats at 8g + 4h) yields k = 8g + 2j + h, j = 0..3: exactly the A/B operand of MFMA step
    // t = 4g + j (lane l holds k = 2t + (l >> 5)); the fmaf chain therefore runs over k ascending.
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            float* dst = As + (p * 32 + srow) * LDK + sg * 8;
            *reinterpret_cast<f32x4*>(dst) = f32x4{ra[p][0][0], ra[p][0][2], ra[p][1][0], ra[p][1][2]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{ra[p][0][1], ra[p][0][3], ra[p][1][1], ra[p][1][3]};
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            float* dst = Bs + (p * 32 + srow) * LDK + sg * 8;
            uint4 w = rb[p];
            *reinterpret_cast<f32x4*>(dst) = f32x4{cz_bits_f32(w.x << 16), cz_bits_f32(w.y << 16), cz_bits_f32(w.z << 16), cz_bits_f32(w.w << 16)};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cz_bits_f32(w.x & 0xffff0000u), cz_bits_f32(w.y & 0xffff0000u),
                                                      cz_bits_f32(w.z & 0xffff0000u), cz_bits_f32(w.w & 0xffff0000u)};
        }
    };

    f32x16 tot[TM][TN];
    const int nk = K / 64;
#ifdef CZ_GEMM_CLOCK
    const bool clk = tid == 0 && (blockIdx.x & 63) == 0;
    unsigned long long ckin = 0, ck0 = 0, rt0 = 0;
    constexpr int CKO = 8 * ((BN == 64 ? 1 : 0) + (EPI == 1 ? 2 : 0));      // [0] BN 128 plain (lm_head), [8] BN 64 (q|k|v, o, down), [16] BN 128 SwiGLU (gate|up)
    if (clk) ckin = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (PIPE) {
    static_assert(!PIPE || (BM % 64 == 0 && BN % 64 == 0), "half-tile staging: 64 rows per pass");
    constexpr int QA = BM / 64, QB = BN / 64;     // staging passes of a half tile (256 threads: 64 rows x 4 quarters of 8 k)
    const int hrow = tid >> 2, hq = tid & 3;
    unsigned aoh[QA];
#pragma unroll
    for (int p = 0; p < QA; ++p) aoh[p] = ((unsigned)min(p * 64 + hrow, mrem) * (unsigned)lda + hq * 8) * 4u;
    const unsigned woh = ((unsigned)hrow * (unsigned)K + hq * 8) * 2u;
    // two register sets: set x is staged (piece by piece, between the MFMA groups of the running half) while set x ^ 1 receives the half after it
    f32x4 ha[2][QA][2];
    uint4 hb[2][QB];
    // float offsets inside a half-tile image: this thread's even-k / odd-k quads of 8-group hq of row hrow (columns 2 hq and 2 hq + 1, swizzled), and this
    // lane's operand quad of group g (column 2 g + lh of row lr, swizzled) — computed once
    const int wsw = (hrow >> 1) & 7, rsw = (lr >> 1) & 7;
    const int wev = hrow * LDH + ((2 * hq) ^ wsw) * 4, wod = hrow * LDH + ((2 * hq + 1) ^ wsw) * 4;
    int rof[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rof[g] = lr * LDH + ((2 * g + lh) ^ rsw) * 4;
    auto load_half = [&](int x, int h) {          // half tile h: k = 32 h .. 32 h + 31
#pragma unroll
        for (int p = 0; p < QA; ++p) {
            ha[x][p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoh[p], h * 128, 0));
            ha[x][p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoh[p], h * 128 + 16, 0));
        }
#pragma unroll
        for (int p = 0; p < QB; ++p) {
            const auto w = __builtin_amdgcn_raw_buffer_load_b128(rsW, woh, (p * 64 * K + h * 32) * 2, 0);
            hb[x][p] = uint4{w[0], w[1], w[2], w[3]};
        }
    };
    // staging pieces 0 .. QA + QB - 1 of register set x into buffer buf: the same even/odd image per group of 8 k as the 64-wide tile
    auto store_piece = [&](int x, int buf, int q) {
        float* Ab = smem + buf * ((BM + BN) * LDH);
        float* Bb = Ab + BM * LDH;
        if (q < QA) {
            const int p = q;
            float* dst = Ab + p * 64 * LDH;                           // (the swizzle of row p * 64 + hrow is that of row hrow)
            *reinterpret_cast<f32x4*>(dst + wev) = f32x4{ha[x][p][0][0], ha[x][p][0][2], ha[x][p][1][0], ha[x][p][1][2]};
            *reinterpret_cast<f32x4*>(dst + wod) = f32x4{ha[x][p][0][1], ha[x][p][0][3], ha[x][p][1][1], ha[x][p][1][3]};
        } else if (q < QA + QB) {
            const int p = q - QA;
            float* dst = Bb + p * 64 * LDH;
            const uint4 w = hb[x][p];
            *reinterpret_cast<f32x4*>(dst + wev) = f32x4{cz_bits_f32(w.x << 16), cz_bits_f32(w.y << 16), cz_bits_f32(w.z << 16), cz_bits_f32(w.w << 16)};
            *reinterpret_cast<f32x4*>(dst + wod) = f32x4{cz_bits_f32(w.x & 0xffff0000u), cz_bits_f32(w.y & 0xffff0000u),
                                                      cz_bits_f32(w.z & 0xffff0000u), cz_bits_f32(w.w & 0xffff0000u)};
        }
    };
    static_assert(QA + QB <= 4, "one staging piece behind each of the four MFMA groups of a half tile");
    f32x16 acc[TM][TN];
    // the half tile in `buf`: four groups of 16 MFMA steps... per accumulator tile 4 steps each (k ascending); behind group g the staging piece g of
    // register set x goes to the OTHER buffer (stage == false: nothing left to stage)
    auto mfma_half = [&](int buf, bool first, bool stage, int x) {
        const float* Ab = smem + buf * ((BM + BN) * LDH);
        const float* Bb = Ab + BM * LDH;
        f32x4 af[2][TM], bf[2][TN];                  // operand fragments of group g and, requested before group g's MFMAs, of group g + 1
        auto frags = [&](int g, int w) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[w][i] = *reinterpret_cast<const f32x4*>(Ab + (wm * TM + i) * 32 * LDH + rof[g]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[w][j] = *reinterpret_cast<const f32x4*>(Bb + (wn * TN + j) * 32 * LDH + rof[g]);
        };
        frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) frags(g + 1, (g + 1) & 1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (first && g == 0 && s4 == 0) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s4], bf[g & 1][j][s4], z, 0, 0, 0);
                        } else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s4], bf[g & 1][j][s4], acc[i][j], 0, 0, 0);
                    }
            if (stage) store_piece(x, buf ^ 1, g);
            __builtin_amdgcn_sched_barrier(0);          // (keeps the piece behind ITS group: the scheduler would gather the pieces in front of the first MFMA)
        }
    };
    const int nh = 2 * nk;
    load_half(0, 0);
#pragma unroll
    for (int q = 0; q < QA + QB; ++q) store_piece(0, 0, q);
    load_half(1, 1);
#ifdef CZ_GEMM_CLOCK
    if (clk) { ck0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int c = 0; c < nk; ++c) {
        // here: buffer 0 holds half 2c (visible after the barrier), register set 1 holds half 2c + 1
        __syncthreads();
        if (2 * c + 2 < nh) load_half(0, 2 * c + 2);
        mfma_half(0, true, true, 1);
        __syncthreads();                                // buffer 1 staged by every wave; buffer 0 read by every wave
        if (2 * c + 3 < nh) load_half(1, 2 * c + 3);
        mfma_half(1, false, 2 * c + 2 < nh, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (c == 0) tot[i][j] = acc[i][j];
                else
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
            }
    }
  } else {
    load_tile(0);
    store_tile();
    __syncthreads();
#ifdef CZ_GEMM_CLOCK
    if (clk) { ck0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long ksec[4] = {0, 0, 0, 0}, kprev = ck0;
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);            // global loads in flight under the MFMA chain
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(As + ((wm * TM + i) * 32 + lr) * LDK + g * 8 + lh * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(Bs + ((wn * TN + j) * 32 + lr) * LDK + g * 8 + lh * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (kt == 0) tot[i][j] = acc[i][j];
                else
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
            }
#ifdef CZ_GEMM_CLOCK
        // (tuning build) where a K tile's cycles go for wave 0: [24] MFMA section, [25] wait at the barrier behind it, [26] staging stores, [27] wait at the second barrier
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        if (clk) s0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (clk) s1 = __builtin_amdgcn_s_memtime();
        if (kt + 1 < nk) { store_tile(); if (clk) s2 = __builtin_amdgcn_s_memtime(); __syncthreads(); }
        if (clk) {
            s3 = __builtin_amdgcn_s_memtime();
            if (CKO == 0) { ksec[0] += s0 - kprev; ksec[1] += s1 - s0; if (kt + 1 < nk) { ksec[2] += s2 - s1; ksec[3] += s3 - s2; } }
            kprev = s3;
        }
#else
        __syncthreads();                                // every wave is done reading this K tile
        if (kt + 1 < nk) { store_tile(); __syncthreads(); }
#endif
    }
#ifdef CZ_GEMM_CLOCK
    if (clk && CKO == 0) for (int q = 0; q < 4; ++q) atomicAdd(&g_gemm_clock[24 + q], ksec[q]);
#endif
  }
#ifdef CZ_GEMM_CLOCK
    unsigned long long ck1 = 0;
    if (clk) {
        ck1 = __builtin_amdgcn_s_memtime(); const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_gemm_clock[CKO + 0], ck1 - ck0); atomicAdd(&g_gemm_clock[CKO + 1], rt1 - rt0); atomicAdd(&g_gemm_clock[CKO + 2], 1ull);
        atomicAdd(&g_gemm_clock[CKO + 3], (unsigned long long)nk); atomicAdd(&g_gemm_clock[CKO + 4], ck0 - ckin);
    }
#define CZ_TAIL_STAMP() do { if (clk) atomicAdd(&g_gemm_clock[CKO + 5], (unsigned long long)__builtin_amdgcn_s_memtime() - ck1); } while (0)
#else
#define CZ_TAIL_STAMP() do { } while (0)
#endif
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int row0 = bm * BM + wm * TM * 32, col0 = bn * BN + wn * TN * 32;      // block- and wave-uniform
    const bool full = bm * BM + BM <= M && !row_mask;                            // every row of the tile is stored: no per-row predicate
    // full tiles: one vector instruction per store / residual load (descriptor at the tile's origin, the lane's fixed byte offset, a scalar offset per row)
    const int wrow = wm * TM * 32, wcol = wn * TN * 32;                          // this wave's corner inside the tile (scalar)
    if (EPI == 1) {
        static_assert(EPI != 1 || TN == 2, "the SwiGLU epilogue pairs the two column tiles of a wave");
        if (full) {
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)(bm * BM) * ldc + bn * BN / 2, 0, 0xffffffffu, 0x00020000);
            const unsigned loff = ((unsigned)(4 * lh) * (unsigned)ldc + lr) * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = cz_siluf(tot[i][0][r]) * tot[i][TN - 1][r];  // silu(gate) * up (llama.rs MLP)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, loff, ((wrow + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + wcol / 2) * 4, 0);
                }
            CZ_TAIL_STAMP();
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int hcol = col0 / 2 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M) C[(size_t)row * ldc + hcol] = cz_siluf(tot[i][0][r]) * tot[i][TN - 1][r];
            }
        }
        CZ_TAIL_STAMP();
        return;
    }
    if (full) {
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)(bm * BM) * ldc + bn * BN, 0, 0xffffffffu, 0x00020000);
        const unsigned loff = ((unsigned)(4 * lh) * (unsigned)ldc + lr) * 4u;
        if (R) {
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + (size_t)(bm * BM) * ldr + bn * BN), 0, 0xffffffffu, 0x00020000);
            const unsigned roff = ((unsigned)(4 * lh) * (unsigned)ldr + lr) * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, roff, ((wrow + i * 32 + (r & 3) + 8 * (r >> 2)) * ldr + wcol + j * 32) * 4, 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[i][j][r] + rv[r]), rsC, loff, ((wrow + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + wcol + j * 32) * 4, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = tot[i][j][r];
                        if (EPI == 2) { const float z = v > 0.0f ? v : 0.0f; v = z * z; }     // RWKV7 channel-mix key: relu(x)^2 (rwkv7.rs:426)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, loff, ((wrow + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + wcol + j * 32) * 4, 0);
                    }
        }
        CZ_TAIL_STAMP();
        return;
    }
    // ragged last row tile, or a row mask: per-row predicates
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 32 + lr;
            // the residual values first, all in flight together (rows past M re-read row M - 1): a load + wait inside a per-element
            // `row < M` branch costs sixteen dependent round trips per tile
            float rv[16];
            if (R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                    rv[r] = R[(size_t)row * ldr + col];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = R ? tot[i][j][r] + rv[r] : tot[i][j][r];
                if (EPI == 2) { const float z = tot[i][j][r] > 0.0f ? tot[i][j][r] : 0.0f; v = z * z; }     // RWKV7 channel-mix key: relu(x)^2 (rwkv7.rs:426)
                if (row < M && (!row_mask || row_mask[row])) C[(size_t)row * ldc + col] = v;       // (row_mask: RWKV literal symbols do not step the model, their logits stay)
            }
        }
    CZ_TAIL_STAMP();
}

// ------------------------------------------------------------------------------------------------------------
// Skinny GEMM, M <= 64 rows.  Block = NCOL(16|32)-column slab of W x all rows; 8 waves; wave w owns K chunks
// w, w+8, ...  Per chunk: A fragment A[i = l&15][k = 4t + (l>>4)] straight from global (tiny, L2-resident),
// B fragment from the bf16 weights W[n = n0 + (l&15)][k], v_mfma_f32_16x16x4_f32 chain t = 0..15 (k ascending).
// Chunk partials go to LDS and are summed in chunk order round by round (8 chunks per round).
template <int MT>   // number of 16-row tiles (1..4)
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const float* __restrict__ A, int lda,
                                                          const uint16_t* __restrict__ W, float* __restrict__ C, int ldc,
                                                          const float* __restrict__ R, int ldr, int M, int N, int K) {
    __shared__ float part[8][MT * 16][16 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int li = lane & 15, lg = lane >> 4;
    const int nch = K / 64;
    // ordered accumulation: thread `tid` owns output elements (row = tid / 16 + 32 * e, col = tid % 16), e = 0, 1
    const int orow = tid >> 4, ocol = tid & 15;
    float total[2] = {0.0f, 0.0f};
    for (int c0 = 0; c0 < nch; c0 += 8) {
        const int c = c0 + wave;
        if (c < nch) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0, 0, 0, 0};
            const uint16_t* wrow = W + (size_t)(n0 + li) * K + c * 64;
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
                float b = cz_bf16_to_f32(wrow[4 * t + lg]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    int row = m * 16 + li;
                    float a = row < M ? A[(size_t)row * lda + c * 64 + 4 * t + lg] : 0.0f;
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
                }
            }
            // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][m * 16 + 4 * lg + r][li] = acc[m][r];
        }
        __syncthreads();
        const int lim = nch - c0 < 8 ? nch - c0 : 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = orow + 32 * e;
            if (row < MT * 16)
                for (int w = 0; w < lim; ++w) {
                    float p = part[w][row][ocol];
                    total[e] = (c0 + w == 0) ? p : total[e] + p;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = orow + 32 * e;
        if (row < MT * 16 && row < M) {
            float v = total[e];
            if (R) v = v + R[(size_t)row * ldr + n0 + ocol];
            C[(size_t)row * ldc + n0 + ocol] = v;
        }
    }
}

int g_gemm_pipe = 3;            // cz_debug_set "gemm_pipe": bit 0 = the 128 x 128 tiles, bit 1 = the 128 x 64 tiles stage half K tiles between their own MFMAs (PIPE); 0 = the one-buffer form
int g_gemm_lds_kb = 0;          // cz_debug_set "gemm_lds_kb": dynamic LDS per block of the big-M GEMM at least this many KB (0 = what the tiles need)
template <int BM, int BN, int WM, int WN, int EPI = 0>
static hipError_t launch_big(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R,
                             int ldr, int M, int N, int K, const unsigned char* row_mask = nullptr) {
    const int nbx = N / BN, nby = (M + BM - 1) / BM, nblocks = nbx * nby;
    if constexpr (BM == 128) {
        if (BN == 128 ? (g_gemm_pipe & 1) : (g_gemm_pipe & 2)) {             // bit 0: the 128 x 128 tiles, bit 1: the 128 x 64 tiles (two blocks per CU instead of three)
            size_t smem = (size_t)2 * (BM + BN) * LDH * sizeof(float);
            if ((size_t)g_gemm_lds_kb * 1024 > smem) smem = (size_t)g_gemm_lds_kb * 1024;
            hipLaunchKernelGGL((gemm_bf16w_kernel<BM, BN, WM, WN, EPI, true>), dim3(nblocks), dim3(256), smem, st, A, lda, W, C, ldc, R, ldr, M, N, K, nbx, nblocks, row_mask);
            return hipGetLastError();
        }
    }
    size_t smem = (size_t)(BM + BN) * LDK * sizeof(float);
    if ((size_t)g_gemm_lds_kb * 1024 > smem) smem = (size_t)g_gemm_lds_kb * 1024;      // tuning knob: fewer GEMM blocks per CU (room for another job's blocks beside them)
    auto kern = gemm_bf16w_kernel<BM, BN, WM, WN, EPI>;        // (its LDS limit is raised by cz_gemm_prepare, once per device at session open)
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, st, A, lda, W, C, ldc, R, ldr, M, N, K, nbx, nblocks, row_mask);
    return hipGetLastError();
}

template <int BM, int BN, int EPI>
static hipError_t big_prepare() {
    if constexpr (BM == 128) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16w_kernel<BM, BN, 2, 2, EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16w_kernel<BM, BN, 2, 2, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // (the tiles need (BM + BN) * LDK * 4; the rest is for the gemm_lds_kb knob)
}
// per device, at session open: dynamic-LDS limits of every instantiation the launchers below use (a lazily set process-wide flag would
// leave a second device's kernels at the 48 KB default and race between the host threads of the decode lanes)
hipError_t cz_gemm_prepare() {
    hipError_t e = big_prepare<64, 128, 1>();
    if (e == hipSuccess) e = big_prepare<128, 128, 1>();
    if (e == hipSuccess) e = big_prepare<64, 64, 0>();
    if (e == hipSuccess) e = big_prepare<128, 128, 0>();
    if (e == hipSuccess) e = big_prepare<128, 64, 0>();
    if (e == hipSuccess) e = big_prepare<64, 64, 2>();
    if (e == hipSuccess) e = big_prepare<128, 128, 2>();
    if (e == hipSuccess) e = big_prepare<128, 64, 2>();
    return e;
}

// Wi[(c / 32) * 64 + (c % 32)] = gate row c, Wi[(c / 32) * 64 + 32 + (c % 32)] = up row c     (W = [gate rows (f) | up rows (f)], each K wide)
__global__ void gu_interleave_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wi, int f, int K) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)2 * f * K; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / K; const int k = (int)(i % K);
        const int up = r >= (size_t)f, c = (int)(up ? r - f : r);
        Wi[((size_t)(c / 32) * 64 + (up ? 32 : 0) + (c % 32)) * K + k] = W[i];
    }
}
hipError_t cz_launch_gu_interleave(hipStream_t st, const uint16_t* W, uint16_t* Wi, int f, int K) {
    if (f % 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gu_interleave_kernel, dim3(1024), dim3(256), 0, st, W, Wi, f, K);
    return hipGetLastError();
}
// H[M][f] = silu(A . Wgate^T) * (A . Wup^T) with Wi from cz_launch_gu_interleave (big-M path only)
// mid-M (one wide lock-step decode group, a few hundred rows): 64-row tiles, so that a projection of a few hundred rows still spreads
// over most of the 256 CUs (a 128-row tile would leave M / 128 x N / 64 = a few dozen blocks)
int g_gemm_mid_rows = 2048;      // tuning knob (cz_debug_set "gemm_mid_rows"): M <= this uses the 64-row tiles
hipError_t cz_launch_gemm_swiglu(hipStream_t st, const float* A, int lda, const uint16_t* Wi, float* H, int ldh, int M, int f, int K) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || (2 * f) % 128 != 0) return hipErrorInvalidValue;
    if (M <= g_gemm_mid_rows) return launch_big<64, 128, 2, 2, 1>(st, A, lda, Wi, H, ldh, nullptr, 0, M, 2 * f, K);
    return launch_big<128, 128, 2, 2, 1>(st, A, lda, Wi, H, ldh, nullptr, 0, M, 2 * f, K);
}

hipError_t cz_launch_gemm(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, const float* R, int ldr,
                          int M, int N, int K, int force_big) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || N % 16 != 0) return hipErrorInvalidValue;
    if (M <= 64 && !force_big) {
        const int mt = (M + 15) / 16;
        dim3 grid(N / 16), block(512);
        switch (mt) {
            case 1: hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, block, 0, st, A, lda, W, C, ldc, R, ldr, M, N, K); break;
            case 2: hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, block, 0, st, A, lda, W, C, ldc, R, ldr, M, N, K); break;
            case 3: hipLaunchKernelGGL(gemm_skinny_kernel<3>, grid, block, 0, st, A, lda, W, C, ldc, R, ldr, M, N, K); break;
            default: hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, block, 0, st, A, lda, W, C, ldc, R, ldr, M, N, K); break;
        }
        return hipGetLastError();
    }
    if (N % 64 != 0) return hipErrorInvalidValue;
    if (M <= g_gemm_mid_rows && (long)((M + 127) / 128) * (N / 64) < 512) return launch_big<64, 64, 2, 2>(st, A, lda, W, C, ldc, R, ldr, M, N, K);
    if (N % 128 == 0) return launch_big<128, 128, 2, 2>(st, A, lda, W, C, ldc, R, ldr, M, N, K);
    return launch_big<128, 64, 2, 2>(st, A, lda, W, C, ldc, R, ldr, M, N, K);
}
// C[M][N] = A . W^T for the rows with row_mask != 0 only (the others keep what C held), tiled kernel, any M
hipError_t cz_launch_gemm_masked(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K, const unsigned char* row_mask) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || N % 64 != 0) return hipErrorInvalidValue;
    if (M <= g_gemm_mid_rows && (long)((M + 127) / 128) * (N / 64) < 512) return launch_big<64, 64, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K, row_mask);
    if (N % 128 == 0) return launch_big<128, 128, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K, row_mask);
    return launch_big<128, 64, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K, row_mask);
}
// C[M][N] = relu(A . W^T)^2 on the tiled kernel, any M (the RWKV7 channel-mix key projection of a sequence-parallel encode)
hipError_t cz_launch_gemm_relusq(hipStream_t st, const float* A, int lda, const uint16_t* W, float* C, int ldc, int M, int N, int K) {
    if (M <= 0) return hipSuccess;
    if (K % 64 != 0 || N % 64 != 0) return hipErrorInvalidValue;
    if (M <= g_gemm_mid_rows && (long)((M + 127) / 128) * (N / 64) < 512) return launch_big<64, 64, 2, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K);
    if (N % 128 == 0) return launch_big<128, 128, 2, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K);
    return launch_big<128, 64, 2, 2, 2>(st, A, lda, W, C, ldc, nullptr, 0, M, N, K);
}

// kernels_attn.hip — causal GQA attention in f32, shared by segment-parallel encode (many rows per sequence) and
// lock-step decode (one row per stream).  Replaces candle_transformers::models::llama CausalSelfAttention
// (third-party; call sites src/models.rs:94, :110):  att = q k^T / sqrt(64) [+ causal mask] ; softmax_last_dim ; att v.
//
// Numerics: score = fmaf chain over d = 0..63 (one canonical chunk) * 0.125 ; row max ; e = cz_expf(s - max);
// sum(e) in canonical 64-chunk order ; p = e / sum ; out[d] = sum_j p_j v_j[d] in canonical order over keys
// (64-key chunks, fmaf chain j ascending inside a chunk, chunk partials added left to right).  Masked keys contribute
// exp(-inf) = 0, which leaves every partial unchanged, so a row's result does not depend on how many later rows
// share its launch: "step == batched".
//
// Block = (sequence, kv head, tile of RT = 16/rep query rows); the rep = n_heads/n_kv query heads of the kv head are
// packed into the 16 MFMA rows ("slots": slot = local_row * rep + head_in_group), so K/V are read once per group.
// 4 waves: QK^T: wave w takes keys 16w..16w+15 of each 64-key chunk (16 x v_mfma_f32_16x16x4_f32, d ascending);
// PV: wave w takes output columns 16w..16w+15 for every chunk in order (16 MFMAs per chunk, keys ascending).
// The whole score tile [16][nk] lives in LDS (<= 16 x 1540 x 4 B = 98.6 KB at nk = 1535).
// Bound: LDS/MFMA latency; algorithmic flops = 4 * 64 * rows * nk_avg per q head.
#include "cz_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define KS 68   // K stage stride (floats)
#define VS 80   // V stage stride (floats)

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ Q, int ldq, float* __restrict__ O, int ldo,
                                                        const cz_seq_desc* __restrict__ descs, int n_heads, int n_kv, int s_stride,
                                                        size_t kv_off, const int* __restrict__ pos_dev) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cz_seq_desc sd = descs[blockIdx.z];
    if (pos_dev) sd.pos0 = *pos_dev;
    sd.k += kv_off; sd.v += kv_off;
    const int rep = n_heads / n_kv, RT = 16 / rep;
    const int tile = blockIdx.x, kvh = blockIdx.y;
    if (tile * RT >= sd.nrows) return;
    const int rows_here = sd.nrows - tile * RT < RT ? sd.nrows - tile * RT : RT;
    const int nslots = rows_here * rep;
    const int nk_blk = sd.pos0 + tile * RT + rows_here;            // keys visible to the last row of the tile
    const int nch = (nk_blk + 63) / 64;

    float* S = smem;                                               // [16][s_stride]
    float* stage = smem + 16 * s_stride;                           // [64][VS]
    float* red = stage + 64 * VS;                                  // [16][32] scratch (max / chunk sums)
    float* rowv = red + 16 * 32;                                   // [16] row max, then row sum

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;

    // slot -> (global row, head, last visible key)
    auto slot_row = [&](int s) { return sd.row0 + tile * RT + s / rep; };
    auto slot_head = [&](int s) { return kvh * rep + s % rep; };
    auto slot_lastkey = [&](int s) { return sd.pos0 + tile * RT + s / rep; };

    // A operand of QK^T: lane (i = li, g = lg) holds Q[slot i][d = 4t + g], t = 0..15
    float qreg[16];
    {
        const bool valid = li < nslots;
        const float* qp = valid ? Q + (size_t)slot_row(li) * ldq + slot_head(li) * 64 : nullptr;
#pragma unroll
        for (int t = 0; t < 16; ++t) qreg[t] = valid ? qp[4 * t + lg] : 0.0f;
    }

    // ---- phase A: scores ----
    for (int c = 0; c < nch; ++c) {
        // stage K chunk [64 keys][64 d] (zeros past nk_blk)
        for (int i = tid; i < 64 * 16; i += 256) {
            const int key = i >> 4, q4 = i & 15;
            const int j = c * 64 + key;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (j < nk_blk) {
                if (sd.kt_stride) {
                    const float* kp = sd.k + (size_t)(kvh * 64 + q4 * 4) * sd.kt_stride + j;
                    v = f32x4{kp[0], kp[sd.kt_stride], kp[2 * (size_t)sd.kt_stride], kp[3 * (size_t)sd.kt_stride]};
                } else v = *reinterpret_cast<const f32x4*>(sd.k + (size_t)j * sd.kv_stride + kvh * 64 + q4 * 4);
            }
            *reinterpret_cast<f32x4*>(stage + key * KS + q4 * 4) = v;
        }
        __syncthreads();
        f32x4 acc = f32x4{0, 0, 0, 0};
        const float* kb = stage + (wave * 16 + li) * KS + lg;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qreg[t], kb[4 * t], acc, 0, 0, 0);
        // C/D: col = li -> key c*64 + wave*16 + li ; row = 4*lg + r -> slot
        const int key = c * 64 + wave * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * lg + r;
            float v = acc[r] * 0.125f;
            if (s >= nslots || key > slot_lastkey(s)) v = -INFINITY;
            S[s * s_stride + key] = v;
        }
        __syncthreads();
    }
    const int nkp = nch * 64;
    // ---- row max (exact in any order) ----
    {
        const int s = tid >> 4, q = tid & 15;
        float m = -INFINITY;
        for (int j = q; j < nkp; j += 16) { float v = S[s * s_stride + j]; m = v > m ? v : m; }
        red[s * 32 + q] = m;
    }
    __syncthreads();
    if (tid < 16) {
        float m = red[tid * 32];
        for (int q = 1; q < 16; ++q) { float v = red[tid * 32 + q]; m = v > m ? v : m; }
        rowv[tid] = m;
    }
    __syncthreads();
    // ---- e = exp(s - max) ----
    for (int i = tid; i < 16 * nkp; i += 256) {
        const int s = i / nkp, j = i % nkp;
        float* p = S + s * s_stride + j;
        *p = s < nslots ? cz_expf(*p - rowv[s]) : 0.0f;
    }
    __syncthreads();
    // ---- canonical sum: (slot, chunk) sequential partials, then chunks left to right ----
    for (int i = tid; i < 16 * nch; i += 256) {
        const int s = i / nch, c = i % nch;
        const float* p = S + s * s_stride + c * 64;
        float a = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) a = a + p[k];
        red[s * 32 + c] = a;
    }
    __syncthreads();
    if (tid < 16) {
        float t = red[tid * 32];
        for (int c = 1; c < nch; ++c) t = t + red[tid * 32 + c];
        rowv[tid] = t;
    }
    __syncthreads();
    for (int i = tid; i < 16 * nkp; i += 256) {
        const int s = i / nkp, j = i % nkp;
        float* p = S + s * s_stride + j;
        if (s < nslots) *p = *p / rowv[s];
    }
    __syncthreads();
    // ---- phase B: out = P V, wave w -> columns 16w.. ----
    f32x4 tot = f32x4{0, 0, 0, 0};
    for (int c = 0; c < nch; ++c) {
        for (int i = tid; i < 64 * 16; i += 256) {
            const int key = i >> 4, q4 = i & 15;
            const int j = c * 64 + key;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (j < nk_blk) v = *reinterpret_cast<const f32x4*>(sd.v + (size_t)j * sd.kv_stride + kvh * 64 + q4 * 4);
            *reinterpret_cast<f32x4*>(stage + key * VS + q4 * 4) = v;
        }
        __syncthreads();
        f32x4 acc = f32x4{0, 0, 0, 0};
        const float* pa = S + li * s_stride + c * 64 + lg;          // A[i = li][k = 4t + lg] = P[slot][key]
        const float* vb = stage + lg * VS + wave * 16 + li;         // B[k = 4t + lg][j = li] = V[key][col]
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * t], vb[4 * t * VS], acc, 0, 0, 0);
        if (c == 0) tot = acc;
        else
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = tot[r] + acc[r];
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s = 4 * lg + r;
        if (s < nslots) O[(size_t)slot_row(s) * ldo + slot_head(s) * 64 + wave * 16 + li] = tot[r];
    }
}


// ------------------------------------------------------------------------------------------------------------
// attention_v2_kernel<REP, NI>: same arithmetic, scores kept in REGISTERS.
// Block = (sequence, kv head, tile of 16 query rows); the REP query heads of the kv group are REP MFMA row tiles that
// share every staged K/V tile.  4 waves: in the QK^T phase wave w owns key tile w of every 64-key chunk and keeps its
// REP x 16x16 score tiles in registers for all NI chunks (NI*REP*4 VGPRs, one wave per SIMD); in the P.V phase wave w owns
// output columns 16w..16w+15 for every chunk in order.  LDS holds only one K/V tile and one e/p tile (35 KB).
// QK^T is computed once; exact two-sweep softmax (max, then e and canonical row sums, then p = e/sum).
#ifdef CZ_ATTN_DBG     // tuning build (make attn_dbg): parts of the kernel switched off by a device-side mask, to time what is left (results are wrong)
__device__ int g_attn_dbg_dev;
extern "C" void cz_debug_attn_dbg(int v) { hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg_dev), &v, sizeof v); }
#define ADBG(bit) (dbg & (bit))
// phase stamps (shader-clock cycles, thread 0 of the blocks (tile & 15) == 15 whose tile has all NI chunks): [0] blocks, [1] Q load .. first K staged, [2] phase 1 (scores),
// [3] phase 2 (row maxima), [4] phase 3 (exp + row sums), [5] sums combined, [6] phase 4 (p, P.V), [7] output stores; [8] chunks
__device__ unsigned long long g_attn_stamp[16];
extern "C" void cz_debug_attn_stamps(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_attn_stamp), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), z, sizeof z); }
}
#define AST_BEGIN const bool ast_on = threadIdx.x == 0 && (blockIdx.x & 15) == 15 && nch == NI; unsigned long long ast_t = ast_on ? __builtin_amdgcn_s_memtime() : 0, ast_t2 = ast_t
#define AST(ph) do { if (ast_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_attn_stamp[ph], t_ - ast_t); ast_t = t_; } } while (0)
#define AST2(ph) do { if (ast_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_attn_stamp[ph], t_ - ast_t2); ast_t2 = t_; } } while (0)
#else
#define ADBG(bit) false
#define AST_BEGIN do { } while (0)
#define AST(ph) do { } while (0)
#endif
#define V2_KS 68
#define V2_VS 80
#define V2_QS 20                        // floats per lane of the LDS copy of the Q operand (16 + 4: the four ds_read_b128 of a wave hit distinct banks)
template <int REP, int NI>
__global__ __launch_bounds__(256, (REP * NI <= 24 ? 3 : REP * NI <= 48 ? 2 : 1)) void attention_v2_kernel(const float* __restrict__ Q, int ldq, float* __restrict__ O, int ldo,
                                                              const cz_seq_desc* __restrict__ descs, size_t kv_off, const int* __restrict__ pos_dev) {
    // DB: K/V tile and e/p tile double-buffered, so that a chunk costs ONE block barrier per phase instead of two (the kernel's waves spend
    // about half their cycles waiting: two waves per SIMD hide little).  Where three blocks share a CU (REP * NI <= 24) the LDS is for them.
    constexpr bool DB = REP * NI > 24 && NI <= 24;      // (NI = 32: the unrolled body outgrows the unroller and the score tiles would land in scratch)
    __shared__ __attribute__((aligned(16))) float stage_[(DB ? 2 : 1) * 64 * V2_VS];   // K tile (stride 68) or V tile (stride 80)
    __shared__ __attribute__((aligned(16))) float Et_[REP * 16 * V2_KS];               // e / p tile of a chunk, [slot][key]
    __shared__ float csum[REP * 16][NI + 1];
    __shared__ float wred[4][REP * 16];
    __shared__ float rsum[REP * 16];
    cz_seq_desc sd = descs[blockIdx.z];
    if (pos_dev) sd.pos0 = *pos_dev;
    sd.k += kv_off; sd.v += kv_off;
    const int tile = blockIdx.x, kvh = blockIdx.y;
    if (tile * 16 >= sd.nrows) return;
    const int rows_here = sd.nrows - tile * 16 < 16 ? sd.nrows - tile * 16 : 16;
    const int nk_blk = sd.pos0 + tile * 16 + rows_here;
    const int nch = (nk_blk + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int row_base = sd.row0 + tile * 16;
#ifdef CZ_ATTN_DBG
    const int dbg = g_attn_dbg_dev;
#endif
    AST_BEGIN;
    // A operand of QK^T: lane (i = li, g = lg) holds Q[row i][head][d = 4t + g].  Where the score tiles alone fill most of the register
    // file (REP * NI >= 48: 192 VGPRs of S) the operand lives in LDS instead — lane-contiguous, so a chunk re-reads it with four
    // ds_read_b128 per head — and the kernel stays under 256 VGPRs without scratch (it spilled 64 registers per lane before).  The
    // 512-key variant of the re-prime (REP * NI = 24) does the same to fit 168 registers: three blocks per CU instead of two.
    constexpr bool QLDS = REP * NI >= 24;
    static_assert(!DB || (QLDS && REP * 64 * V2_QS >= REP * 16 * V2_KS), "the second e/p tile lives where the Q operand was (dead after phase 1)");
    __shared__ __attribute__((aligned(16))) float Qs[QLDS ? REP * 64 * V2_QS : 4];
    auto stage_of = [&](int c) { return stage_ + (DB ? (c & 1) * (64 * V2_VS) : 0); };
    auto et_of = [&](int c) { return (DB && (c & 1)) ? Qs : Et_; };
    float qreg[QLDS ? 1 : REP][16];
    // (unconditional loads from a clamped row, the select afterwards: a load inside a per-element `valid` branch is waited for on the
    // spot, i.e. REP x 16 dependent round trips before the block's first MFMA)
    if (!QLDS || wave == 0) {
        const bool valid = li < rows_here;
#pragma unroll
        for (int rt = 0; rt < REP; ++rt) {
            const float* qp = Q + (size_t)(row_base + (valid ? li : 0)) * ldq + (kvh * REP + rt) * 64 + lg;
            float raw[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) raw[t] = qp[4 * t];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float v = valid ? raw[t] : 0.0f;
                if (QLDS) Qs[(rt * 64 + lane) * V2_QS + t] = v; else qreg[rt][t] = v;
            }
        }
    }
    f32x4 S[NI][REP];
    // K / V chunk loads: one block per CU and one wave per SIMD leave nothing else to hide a miss behind, so the NEXT chunk's four
    // 16-B pieces per thread are fetched into registers while the current chunk's MFMAs run
    // K (row layout) and V chunk loads: descriptor at the sequence's rows of this KV head, a per-lane byte offset inside the chunk (one v_add per pass) clamped by
    // ONE v_min to the chunk's last valid row, the chunk's base as a scalar offset — no 64-bit address arithmetic, no zero-fill (round 4: a vector instruction
    // of this kernel costs 20-35 cycles beside the other block's MFMA bursts; clamps, address arithmetic and zero-fills were ~65 of a score chunk's ~130).
    // Rows past nk_blk re-read the last valid row: their scores are masked to -inf below (key > every row's position), so their p is exactly 0 and
    // fma(0, v, acc) == acc for the finite v of a valid row — the same bits as the zero padding of the definition.
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sd.k + kvh * 64), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sd.v + kvh * 64), 0, 0xffffffffu, 0x00020000);
    const unsigned kv_lane_off_ = ((unsigned)(tid >> 4) * (unsigned)sd.kv_stride + (tid & 15) * 4) * 4u;
    auto load_rows = [&](const __amdgpu_buffer_rsrc_t rs, int c, f32x4* out) {
        unsigned t = tid, kv_lane_off = kv_lane_off_;
        if (REP * NI <= 24) {                                          // the 168-register variants: re-derive the lane's offsets per call (five VALU) instead of keeping them live (they spilled)
            asm volatile("" : "+v"(t));
            kv_lane_off = ((t >> 4) * (unsigned)sd.kv_stride + (t & 15) * 4) * 4u;
        }
        const unsigned lim = (unsigned)(nk_blk - 1 - c * 64) * (unsigned)sd.kv_stride * 4u + (t & 15) * 16u;     // this lane's piece of the chunk's last valid row
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned off = min(kv_lane_off + (unsigned)(16 * p) * (unsigned)sd.kv_stride * 4u, lim);
            out[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, c * 64 * sd.kv_stride * 4, 0));
        }
    };
    auto load_k = [&](int c, f32x4* out) {
        if (ADBG(32)) { out[0] = out[1] = out[2] = out[3] = f32x4{0.5f, 0.25f, 0.125f, 1.0f}; return; }
        if (!sd.kt_stride) { load_rows(rsK, c, out); return; }
#pragma unroll
        for (int p = 0; p < 4; ++p) {                                  // the decode cache's transposed K (re-prime): k[(kvh*64 + d) * kt_stride + j]
            const int key = (tid >> 4) + 16 * p, q4 = tid & 15;
            const int j = c * 64 + key;
            const int jj = j < nk_blk ? j : nk_blk - 1;               // clamp (never a conditional load)
            const float* kp = sd.k + (size_t)(kvh * 64 + q4 * 4) * sd.kt_stride + jj;
            out[p] = f32x4{kp[0], kp[sd.kt_stride], kp[2 * (size_t)sd.kt_stride], kp[3 * (size_t)sd.kt_stride]};
        }
    };
    auto load_v = [&](int c, f32x4* out) {
        if (ADBG(32)) { out[0] = out[1] = out[2] = out[3] = f32x4{0.5f, 0.25f, 0.125f, 1.0f}; return; }
        load_rows(rsV, c, out);
    };
    f32x4 pre[4];
    load_k(0, pre);
    const int klim0 = sd.pos0 + tile * 16 + 4 * lg;                    // the last key this lane's query row r may see is klim0 + r
    // e / p tiles: key k of the chunk lives at column (k & 3) * 16 + (k >> 2) of its row, so that the A operand of MFMA step t — key 4t + lg — is column
    // lg * 16 + t: a lane reads its sixteen operands as FOUR ds_read_b128 instead of eight ds_read2_b32, each of which the two MFMAs behind it waited for
    const int kperm = (li & 3) * 16 + wave * 4 + (li >> 2);            // this lane's key wave * 16 + li
    AST(1);
#ifdef CZ_ATTN_DBG
    ast_t2 = ast_t;
#endif
    // ---- phase 1: scores
#pragma unroll
    for (int c = 0; c < NI; ++c) {
#pragma unroll
        for (int rt = 0; rt < REP; ++rt) S[c][rt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (c < nch) {
            float* stage = stage_of(c);
#ifdef CZ_ATTN_DBG
            if (ast_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            AST2(9);                                                   // (tuning build) [9] wait for this chunk's K in registers, [10] stage it + barrier, [11] the rest of the chunk
#endif
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(stage + ((tid >> 4) + 16 * p) * V2_KS + (tid & 15) * 4) = pre[p];
            __syncthreads();
#ifdef CZ_ATTN_DBG
            AST2(10);
#endif
            if (c + 1 < nch) load_k(c + 1, pre);                       // in flight during the MFMAs below
            float kb[16];
            const float* kbp = stage + (wave * 16 + li) * V2_KS + lg;
            if (!QLDS) {
#pragma unroll
                for (int t = 0; t < 16; ++t) kb[t] = kbp[4 * t];
            }
            const int key = c * 64 + wave * 16 + li, dk = key - klim0;
#pragma unroll
            for (int rt = 0; rt < REP; ++rt) {
                f32x4 acc = f32x4{0, 0, 0, 0};
                if (ADBG(1)) { acc[0] = (float)key; }
                else if (QLDS) {                                            // operands in pieces of four k-steps: 8 live registers instead of 32
                    const f32x4* qp4 = reinterpret_cast<const f32x4*>(Qs + (rt * 64 + lane) * V2_QS);
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const f32x4 q = qp4[t4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[t], kbp[4 * (4 * t4 + t)], acc, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qreg[rt][t], kb[t], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                               // C/D: col = li -> key ; row = 4*lg + r -> query row
                    float v = acc[r] * 0.125f;
                    if (dk > r) v = -INFINITY;                              // causal mask: key > klim0 + r (rows past rows_here: zero queries, never stored — no mask of their own)
                    S[c][rt][r] = v;                                        // (a block-uniform branch that skips the mask off the diagonal costs 430 spilled registers: the score tiles cross it)
                }
            }
            if (!DB) __syncthreads();                                   // (DB: the next chunk goes to the other buffer; the one after that is past the next barrier)
#ifdef CZ_ATTN_DBG
            AST2(11);
#endif
        }
    }
    load_v(0, pre);                                                    // first V chunk: in flight during the softmax phases
    AST(2);
    // ---- phase 2: row max (exact in any order)
    float gmax[REP][4];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < NI; ++c) m = S[c][rt][r] > m ? S[c][rt][r] : m;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const float y = __shfl_xor(m, o); m = y > m ? y : m; }
            if (li == 0) wred[wave][rt * 16 + 4 * lg + r] = m;
        }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = rt * 16 + 4 * lg + r;
            float m = wred[0][s];
#pragma unroll
            for (int w = 1; w < 4; ++w) m = wred[w][s] > m ? wred[w][s] : m;
            gmax[rt][r] = m;
        }
    AST(3);
    // ---- phase 3: e = exp(s - max); canonical row sums (sequential inside a 64-key chunk, chunks left to right)
    // canonical chunk sum of a head's 16 rows (acc = acc + e_k, k ascending from 0) on the MFMA: with B = 1 the chain fma(e_k, 1, acc) rounds
    // e_k + acc once — the same bits as the add — and sixteen 16x16x4 steps replace 64 dependent LDS reads + adds
    auto exp_elem = [&](int c, int rt, int r, float* Et) {
        const int qr = 4 * lg + r;
        const float e = ADBG(2) ? S[c][rt][r] - gmax[rt][r] : cz_expf(S[c][rt][r] - gmax[rt][r]);      // (rows past rows_here: zero queries, finite values, never stored)
        S[c][rt][r] = e;
        Et[(rt * 16 + qr) * V2_KS + kperm] = e;
    };
    // (Issuing chunk c's sixteen dependent sum MFMAs between the exponentials of chunk c + 1 was tried: the longer live ranges cost 24
    // spilled registers at <3,16> and the kernel got 6 % slower, call27.)
    {
#pragma unroll
        for (int c = 0; c < NI; ++c) {
            if (c < nch) {
                float* Et = et_of(c);
#pragma unroll
                for (int rt = 0; rt < REP; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) exp_elem(c, rt, r, Et);
                __syncthreads();
                if (wave < REP && !ADBG(4)) {
                    const f32x4* pa4 = reinterpret_cast<const f32x4*>(Et + (wave * 16 + li) * V2_KS + lg * 16);      // A[i = li][k = 4t + lg] = e[row][key]: column lg * 16 + t
                    f32x4 sacc = f32x4{0, 0, 0, 0};
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const f32x4 a = pa4[t4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], 1.0f, sacc, 0, 0, 0);
                    }
                    if (li == 0) {                                                    // C/D: row = 4*lg + r; every column holds the same sum
#pragma unroll
                        for (int r = 0; r < 4; ++r) csum[wave * 16 + 4 * lg + r][c] = sacc[r];
                    }
                }
                if (!DB) __syncthreads();
            }
        }
        if (DB) __syncthreads();                                        // csum complete
    }
    AST(4);
    if (tid < REP * 16) {
        float t = csum[tid][0];
        for (int c = 1; c < nch; ++c) t = t + csum[tid][c];
        rsum[tid] = t;
    }
    __syncthreads();
    AST(5);
    float rs[REP][4];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[rt][r] = rsum[rt * 16 + 4 * lg + r];
    // ---- phase 4: p = e / sum ; out = P.V (wave w -> output columns 16w.., chunks in order)
    f32x4 tot[REP];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt) tot[rt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        if (c < nch) {
            float* Et = et_of(c);
            float* stage = stage_of(c);
#pragma unroll
            for (int rt = 0; rt < REP; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qr = 4 * lg + r;
                    Et[(rt * 16 + qr) * V2_KS + kperm] = ADBG(8) ? S[c][rt][r] : S[c][rt][r] / rs[rt][r];
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(stage + ((tid >> 4) + 16 * p) * V2_VS + (tid & 15) * 4) = pre[p];
            __syncthreads();
            if (c + 1 < nch) load_v(c + 1, pre);
            float vb[16];
            const float* vbp = stage + lg * V2_VS + wave * 16 + li;          // B[k = 4t + lg][j = li] = V[key][col]
            if (!QLDS) {
#pragma unroll
                for (int t = 0; t < 16; ++t) vb[t] = vbp[4 * t * V2_VS];
            }
#pragma unroll
            for (int rt = 0; rt < REP; ++rt) {
                const f32x4* pa4 = reinterpret_cast<const f32x4*>(Et + (rt * 16 + li) * V2_KS + lg * 16);      // A[i = li][k = 4t + lg] = P[row][key]: column lg * 16 + t
                f32x4 acc = f32x4{0, 0, 0, 0};
                if (ADBG(16)) acc[0] = pa4[0][0];
                else
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const f32x4 a = pa4[t4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], QLDS ? vbp[4 * (4 * t4 + t) * V2_VS] : vb[4 * t4 + t], acc, 0, 0, 0);
                }
                if (c == 0) tot[rt] = acc;
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) tot[rt][r] = tot[rt][r] + acc[r];
            }
            if (!DB) __syncthreads();
        }
    }
    AST(6);
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = 4 * lg + r;
            if (qr < rows_here) O[(size_t)(row_base + qr) * ldo + (kvh * REP + rt) * 64 + wave * 16 + li] = tot[rt][r];
        }
    AST(7);
#ifdef CZ_ATTN_DBG
    if (ast_on) { atomicAdd(&g_attn_stamp[0], 1ull); atomicAdd(&g_attn_stamp[8], (unsigned long long)nch); }
#endif
}
// ------------------------------------------------------------------------------------------------------------
// attention_v3_kernel<REP>: the same arithmetic for ANY context length (re-prime intervals past 1536 tokens: a segment of up to
// max_position_embeddings rows).  v2 keeps the score tiles of all chunks in registers (<= 24 chunks at REP 3), v1 keeps them in LDS
// (<= 2048 keys); here nothing is kept: QK^T is evaluated three times, chunk by chunk — pass 1 the row maxima, pass 2 e = exp(s - max)
// and the canonical row sums (chunk sums on the MFMA, added left to right as they come), pass 3 p = e / sum and P.V.  Every pass
// recomputes bit-identical scores, so the result equals v2's wherever both run (tests/test_gpu_kernels.py).  LDS: one K tile, one V tile,
// one e/p tile, the Q operand (65 KB at REP 3).
template <int REP>
__global__ __launch_bounds__(256, 2) void attention_v3_kernel(const float* __restrict__ Q, int ldq, float* __restrict__ O, int ldo,
                                                              const cz_seq_desc* __restrict__ descs, size_t kv_off, const int* __restrict__ pos_dev) {
    __shared__ __attribute__((aligned(16))) float Ks_[64 * V2_KS];
    __shared__ __attribute__((aligned(16))) float Vs_[64 * V2_VS];
    __shared__ __attribute__((aligned(16))) float Et_[REP * 16 * V2_KS];
    __shared__ __attribute__((aligned(16))) float Qs[REP * 64 * V2_QS];
    __shared__ float wred[4][REP * 16];
    __shared__ float rsum[REP * 16];
    cz_seq_desc sd = descs[blockIdx.z];
    if (pos_dev) sd.pos0 = *pos_dev;
    sd.k += kv_off; sd.v += kv_off;
    const int tile = blockIdx.x, kvh = blockIdx.y;
    if (tile * 16 >= sd.nrows) return;
    const int rows_here = sd.nrows - tile * 16 < 16 ? sd.nrows - tile * 16 : 16;
    const int nk_blk = sd.pos0 + tile * 16 + rows_here;
    const int nch = (nk_blk + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int row_base = sd.row0 + tile * 16;
    if (wave == 0) {                                   // Q operand, lane-contiguous (as v2's QLDS form); clamped loads, select afterwards
        const bool valid = li < rows_here;
#pragma unroll
        for (int rt = 0; rt < REP; ++rt) {
            const float* qp = Q + (size_t)(row_base + (valid ? li : 0)) * ldq + (kvh * REP + rt) * 64 + lg;
            float raw[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) raw[t] = qp[4 * t];
#pragma unroll
            for (int t = 0; t < 16; ++t) Qs[(rt * 64 + lane) * V2_QS + t] = valid ? raw[t] : 0.0f;
        }
    }
    auto load_kv = [&](const float* base, int kt_stride, int c, f32x4* out) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int key = (tid >> 4) + 16 * p, q4 = tid & 15;
            const int j = c * 64 + key;
            const int jj = j < nk_blk ? j : nk_blk - 1;               // clamp (never a conditional load), zero afterwards
            f32x4 v;
            if (kt_stride) {
                const float* kp = base + (size_t)(kvh * 64 + q4 * 4) * kt_stride + jj;
                v = f32x4{kp[0], kp[kt_stride], kp[2 * (size_t)kt_stride], kp[3 * (size_t)kt_stride]};
            } else v = *reinterpret_cast<const f32x4*>(base + (size_t)jj * sd.kv_stride + kvh * 64 + q4 * 4);
            if (j >= nk_blk) v = f32x4{0, 0, 0, 0};
            out[p] = v;
        }
    };
    // scores of chunk c (K tile already in Ks_, behind a barrier): sc[rt][r] = row 4 lg + r, key c*64 + wave*16 + li
    auto scores = [&](int c, f32x4* sc) {
        const float* kbp = Ks_ + (wave * 16 + li) * V2_KS + lg;
        const int key = c * 64 + wave * 16 + li;
#pragma unroll
        for (int rt = 0; rt < REP; ++rt) {
            f32x4 acc = f32x4{0, 0, 0, 0};
            const f32x4* qp4 = reinterpret_cast<const f32x4*>(Qs + (rt * 64 + lane) * V2_QS);
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const f32x4 q = qp4[t4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[t], kbp[4 * (4 * t4 + t)], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = 4 * lg + r;
                float v = acc[r] * 0.125f;
                if (qr >= rows_here || key > sd.pos0 + tile * 16 + qr) v = -INFINITY;
                sc[rt][r] = v;
            }
        }
    };
    auto store_k = [&](const f32x4* pre) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Ks_ + ((tid >> 4) + 16 * p) * V2_KS + (tid & 15) * 4) = pre[p];
    };
    f32x4 prek[4], prev[4];
    // ---- pass 1: row maxima (exact in any order)
    float gmax[REP][4];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) gmax[rt][r] = -INFINITY;
    load_kv(sd.k, sd.kt_stride, 0, prek);
    for (int c = 0; c < nch; ++c) {
        store_k(prek);
        __syncthreads();
        if (c + 1 < nch) load_kv(sd.k, sd.kt_stride, c + 1, prek);
        f32x4 sc[REP];
        scores(c, sc);
#pragma unroll
        for (int rt = 0; rt < REP; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gmax[rt][r] = sc[rt][r] > gmax[rt][r] ? sc[rt][r] : gmax[rt][r];
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = gmax[rt][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const float y = __shfl_xor(m, o); m = y > m ? y : m; }
            if (li == 0) wred[wave][rt * 16 + 4 * lg + r] = m;
        }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = rt * 16 + 4 * lg + r;
            float m = wred[0][s];
#pragma unroll
            for (int w = 1; w < 4; ++w) m = wred[w][s] > m ? wred[w][s] : m;
            gmax[rt][r] = m;
        }
    // ---- pass 2: e = exp(s - max); canonical row sums: the chunk sum on the MFMA (B = 1: fma(e_k, 1, acc) == acc + e_k), chunks left to right
    f32x4 run = f32x4{0, 0, 0, 0};                      // wave < REP: the running row sums of head `wave` (rows 4 lg + r)
    load_kv(sd.k, sd.kt_stride, 0, prek);
    for (int c = 0; c < nch; ++c) {
        store_k(prek);
        __syncthreads();
        if (c + 1 < nch) load_kv(sd.k, sd.kt_stride, c + 1, prek);
        f32x4 sc[REP];
        scores(c, sc);
#pragma unroll
        for (int rt = 0; rt < REP; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = 4 * lg + r;
                Et_[(rt * 16 + qr) * V2_KS + wave * 16 + li] = qr < rows_here ? cz_expf(sc[rt][r] - gmax[rt][r]) : 0.0f;
            }
        __syncthreads();
        if (wave < REP) {
            const float* pa = Et_ + (wave * 16 + li) * V2_KS + lg;
            f32x4 sacc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * t], 1.0f, sacc, 0, 0, 0);
            if (c == 0) run = sacc;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) run[r] = run[r] + sacc[r];
        }
    }
    if (wave < REP && li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rsum[wave * 16 + 4 * lg + r] = run[r];
    }
    __syncthreads();
    float rs[REP][4];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[rt][r] = rsum[rt * 16 + 4 * lg + r];
    // ---- pass 3: p = e / sum ; out = P.V (wave w -> output columns 16w.., chunks in order)
    f32x4 tot[REP];
#pragma unroll
    for (int rt = 0; rt < REP; ++rt) tot[rt] = f32x4{0, 0, 0, 0};
    load_kv(sd.k, sd.kt_stride, 0, prek);
    load_kv(sd.v, 0, 0, prev);
    for (int c = 0; c < nch; ++c) {
        store_k(prek);
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(Vs_ + ((tid >> 4) + 16 * p) * V2_VS + (tid & 15) * 4) = prev[p];
        __syncthreads();
        if (c + 1 < nch) { load_kv(sd.k, sd.kt_stride, c + 1, prek); load_kv(sd.v, 0, c + 1, prev); }
        f32x4 sc[REP];
        scores(c, sc);
#pragma unroll
        for (int rt = 0; rt < REP; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = 4 * lg + r;
                Et_[(rt * 16 + qr) * V2_KS + wave * 16 + li] = qr < rows_here ? cz_expf(sc[rt][r] - gmax[rt][r]) / rs[rt][r] : 0.0f;
            }
        __syncthreads();
        const float* vbp = Vs_ + lg * V2_VS + wave * 16 + li;
#pragma unroll
        for (int rt = 0; rt < REP; ++rt) {
            const float* pa = Et_ + (rt * 16 + li) * V2_KS + lg;
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * t], vbp[4 * t * V2_VS], acc, 0, 0, 0);
            if (c == 0) tot[rt] = acc;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) tot[rt][r] = tot[rt][r] + acc[r];
        }
        __syncthreads();                                   // Ks_ / Vs_ / Et_ are free for the next chunk
    }
#pragma unroll
    for (int rt = 0; rt < REP; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qr = 4 * lg + r;
            if (qr < rows_here) O[(size_t)(row_base + qr) * ldo + (kvh * REP + rt) * 64 + wave * 16 + li] = tot[rt][r];
        }
}
template <int REP>
static hipError_t launch_v3(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nseq, int max_rows,
                            int n_kv, size_t kv_off, const int* pos_dev) {
    const dim3 grid((max_rows + 15) / 16, n_kv, nseq), block(256);
    hipLaunchKernelGGL((attention_v3_kernel<REP>), grid, block, 0, st, Q, ldq, O, ldo, descs, kv_off, pos_dev);
    return hipGetLastError();
}
template <int REP>
static hipError_t launch_v2(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nseq, int max_rows, int max_nk,
                            int n_kv, size_t kv_off, const int* pos_dev) {
    const dim3 grid((max_rows + 15) / 16, n_kv, nseq), block(256);
    const int ni = (max_nk + 63) / 64;
    if (ni <= 8) hipLaunchKernelGGL((attention_v2_kernel<REP, 8>), grid, block, 0, st, Q, ldq, O, ldo, descs, kv_off, pos_dev);
    else if (ni <= 16) hipLaunchKernelGGL((attention_v2_kernel<REP, 16>), grid, block, 0, st, Q, ldq, O, ldo, descs, kv_off, pos_dev);
    else if (ni <= 24) hipLaunchKernelGGL((attention_v2_kernel<REP, 24>), grid, block, 0, st, Q, ldq, O, ldo, descs, kv_off, pos_dev);
    else hipLaunchKernelGGL((attention_v2_kernel<REP, 32>), grid, block, 0, st, Q, ldq, O, ldo, descs, kv_off, pos_dev);
    return hipGetLastError();
}

int g_attn_v1 = 0;     // test knob: 1 = the LDS-resident v1 kernel, 2 = the three-pass v3 kernel wherever it can run (rep <= 4)
static int s_stride_for(int max_nk) { return ((max_nk + 63) / 64) * 64 + 4; }
size_t cz_attention_smem_bytes(int max_nk) { return (size_t)(16 * s_stride_for(max_nk) + 64 * VS + 16 * 32 + 16) * sizeof(float); }
hipError_t cz_attention_prepare(int max_nk) {
    if (max_nk > 2048) max_nk = 2048;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)cz_attention_smem_bytes(max_nk));
}
hipError_t cz_launch_attention(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nseq,
                               int max_rows, int max_nk, int n_heads, int n_kv, size_t kv_off, const int* pos_dev) {
    if (nseq <= 0 || max_rows <= 0) return hipSuccess;
    const int rep = n_heads / n_kv;
    if (rep < 1 || rep > 16 || n_heads % n_kv) return hipErrorInvalidValue;
    // v2 keeps NI*REP score tiles in registers: beyond these shapes the compiler spills them, so v1 (LDS-resident) takes over
    const int ni_ = (max_nk + 63) / 64;
    const bool v2_fits = (rep <= 2 && ni_ <= 32) || (rep == 3 && ni_ <= 24) || (rep == 4 && ni_ <= 16);
    if (g_attn_v1 == 0 && v2_fits && max_rows > 1) {
        switch (rep) {
            case 1: return launch_v2<1>(st, Q, ldq, O, ldo, descs, nseq, max_rows, max_nk, n_kv, kv_off, pos_dev);
            case 2: return launch_v2<2>(st, Q, ldq, O, ldo, descs, nseq, max_rows, max_nk, n_kv, kv_off, pos_dev);
            case 3: return launch_v2<3>(st, Q, ldq, O, ldo, descs, nseq, max_rows, max_nk, n_kv, kv_off, pos_dev);
            default: return launch_v2<4>(st, Q, ldq, O, ldo, descs, nseq, max_rows, max_nk, n_kv, kv_off, pos_dev);
        }
    }
    if (g_attn_v1 != 1 && rep <= 4 && (max_rows > 1 || max_nk > 2048 || g_attn_v1 == 2)) {      // contexts past what v2's registers (and v1's LDS) hold: the three-pass kernel
        switch (rep) {
            case 1: return launch_v3<1>(st, Q, ldq, O, ldo, descs, nseq, max_rows, n_kv, kv_off, pos_dev);
            case 2: return launch_v3<2>(st, Q, ldq, O, ldo, descs, nseq, max_rows, n_kv, kv_off, pos_dev);
            case 3: return launch_v3<3>(st, Q, ldq, O, ldo, descs, nseq, max_rows, n_kv, kv_off, pos_dev);
            default: return launch_v3<4>(st, Q, ldq, O, ldo, descs, nseq, max_rows, n_kv, kv_off, pos_dev);
        }
    }
    if (max_nk > 2048) return hipErrorInvalidValue;         // v1 keeps a [16][nk] score tile in LDS
    const int RT = 16 / rep;
    const int ss = s_stride_for(max_nk);
    dim3 grid((max_rows + RT - 1) / RT, n_kv, nseq);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), cz_attention_smem_bytes(max_nk), st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev);
    return hipGetLastError();
}

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

static int s_stride_for(int max_nk) { return ((max_nk + 63) / 64) * 64 + 4; }
size_t cz_attention_smem_bytes(int max_nk) { return (size_t)(16 * s_stride_for(max_nk) + 64 * VS + 16 * 32 + 16) * sizeof(float); }
hipError_t cz_attention_prepare(int max_nk) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)cz_attention_smem_bytes(max_nk));
}
hipError_t cz_launch_attention(hipStream_t st, const float* Q, int ldq, float* O, int ldo, const cz_seq_desc* descs, int nseq,
                               int max_rows, int max_nk, int n_heads, int n_kv, size_t kv_off, const int* pos_dev) {
    if (nseq <= 0 || max_rows <= 0) return hipSuccess;
    const int rep = n_heads / n_kv;
    if (rep < 1 || rep > 16 || n_heads % n_kv) return hipErrorInvalidValue;
    const int RT = 16 / rep;
    const int ss = s_stride_for(max_nk);
    dim3 grid((max_rows + RT - 1) / RT, n_kv, nseq);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), cz_attention_smem_bytes(max_nk), st, Q, ldq, O, ldo, descs, n_heads, n_kv, ss, kv_off, pos_dev);
    return hipGetLastError();
}

// kernels_misc.hip — memory-bound element/row kernels of the SmolLM2 (LLaMA) block: embedding gather, RMSNorm,
// RoPE (+ K/V placement), SwiGLU.  All are HBM-bound streaming kernels (roofline: 8 TB/s peak): 16-byte accesses,
// no transcendental on the hot path except SwiGLU's owned expf.  They replace candle ops called from
// candle_transformers::models::llama (third-party; call sites src/models.rs:94, :110).
#include "cz_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// embedding gather: X[m][:] = widen(E[tok[m]][:])        (llama.rs `wte.forward`)
__global__ void embed_kernel(const uint32_t* __restrict__ tok, const uint16_t* __restrict__ E, float* __restrict__ X, int M, int d) {
    const int per_row = d / 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / per_row), c = (int)(i % per_row);
        const uint4 w = *reinterpret_cast<const uint4*>(E + (size_t)tok[m] * d + c * 8);
        float* dst = X + (size_t)m * d + c * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{cz_bits_f32(w.x << 16), cz_bits_f32(w.x & 0xffff0000u), cz_bits_f32(w.y << 16), cz_bits_f32(w.y & 0xffff0000u)};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cz_bits_f32(w.z << 16), cz_bits_f32(w.z & 0xffff0000u), cz_bits_f32(w.w << 16), cz_bits_f32(w.w & 0xffff0000u)};
    }
}
hipError_t cz_launch_embed(hipStream_t st, const uint32_t* tok, const uint16_t* E, float* X, int M, int d) {
    if (M <= 0) return hipSuccess;
    long n = (long)M * (d / 8);
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_kernel, dim3(blocks), dim3(256), 0, st, tok, E, X, M, d);
    return hipGetLastError();
}

// RMSNorm (candle_nn::ops::rms_norm CPU kernel): m = sqrt(sum(x*x)/d + eps); y = x / m * w.
// Canonical sum: squares rounded, 64-chunk sequential sums, chunk partials left to right.
// Block = 256 threads = 16 rows x 16 chunk-lanes (d <= 1024).  row_index (optional) gathers source rows.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ X, int ldx, const uint16_t* __restrict__ w,
                                                      float* __restrict__ Y, int ldy, int M, int d, float eps,
                                                      const int* __restrict__ row_index) {
    __shared__ float part[16][17];
    __shared__ float inv_m[16];
    const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
    const int m = blockIdx.x * 16 + r;
    const int nch = d / 64;
    const float* x = nullptr;
    if (m < M) x = X + (size_t)(row_index ? row_index[m] : m) * ldx;
    if (x && c < nch) {
        float p = 0.0f;
        const f32x4* src = reinterpret_cast<const f32x4*>(x + c * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x4 v = src[k];
            float s0 = v[0] * v[0], s1 = v[1] * v[1], s2 = v[2] * v[2], s3 = v[3] * v[3];
            p = p + s0; p = p + s1; p = p + s2; p = p + s3;
        }
        part[r][c] = p;
    }
    __syncthreads();
    if (x && c == 0) {
        float t = part[r][0];
        for (int k = 1; k < nch; ++k) t = t + part[r][k];
        inv_m[r] = sqrtf(t / (float)d + eps);
    }
    __syncthreads();
    // normalise: 16 rows x d elements, coalesced over the block
    const int per_row = d / 4;
    for (int i = tid; i < 16 * per_row; i += 256) {
        const int rr = i / per_row, cc = i % per_row;
        const int mm = blockIdx.x * 16 + rr;
        if (mm >= M) break;
        const float* xs = X + (size_t)(row_index ? row_index[mm] : mm) * ldx;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xs + cc * 4);
        const uint2 wb = *reinterpret_cast<const uint2*>(w + cc * 4);
        const float mval = inv_m[rr];
        f32x4 o;
        o[0] = v[0] / mval * cz_bits_f32(wb.x << 16);
        o[1] = v[1] / mval * cz_bits_f32(wb.x & 0xffff0000u);
        o[2] = v[2] / mval * cz_bits_f32(wb.y << 16);
        o[3] = v[3] / mval * cz_bits_f32(wb.y & 0xffff0000u);
        *reinterpret_cast<f32x4*>(Y + (size_t)mm * ldy + cc * 4) = o;
    }
}
// The same for a FEW HUNDRED rows (one wide lock-step decode group): 4 rows per block instead of 16, so that 512 rows are 128 blocks
// rather than 32, and the rows pass through LDS once (one global round trip: the kernel is pure latency at this size).
#define RS_ROWS 4
#define RS_CH 65                       // floats per 64-chunk in LDS (+1: the nine chain lanes of a row hit different banks)
#define RS_ROW (16 * RS_CH + 9)        // row stride: 1049 = 25 mod 32
__global__ __launch_bounds__(256) void rmsnorm_small_kernel(const float* __restrict__ X, int ldx, const uint16_t* __restrict__ w,
                                                            float* __restrict__ Y, int ldy, int M, int d, float eps) {
    __shared__ float xs[RS_ROWS * RS_ROW];
    __shared__ float part[RS_ROWS][17];
    __shared__ float mv[RS_ROWS];
    // wave r of the block owns row m0 + r: no index arithmetic with divisions, and every global operand — the row AND the weights the
    // last phase needs — is requested before the first barrier (this kernel sits 61 times in the dependent chain of a wide decode step)
    const int tid = threadIdx.x, lane = tid & 63, r = tid >> 6, per_row = d / 4, nch = d / 64, m0 = blockIdx.x * RS_ROWS;
    const int row = m0 + r < M ? m0 + r : M - 1;                     // (clamped address, never a conditional load; the store below is guarded)
    const bool live = m0 + r < M;
    f32x4 v[4]; uint2 wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = lane + 64 * j, ca = cc < per_row ? cc : per_row - 1;
        v[j] = *reinterpret_cast<const f32x4*>(X + (size_t)row * ldx + ca * 4);
        wb[j] = *reinterpret_cast<const uint2*>(w + ca * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = lane + 64 * j;
        if (cc < per_row) {
            float* dst = xs + r * RS_ROW + (cc >> 4) * RS_CH + (cc & 15) * 4;
            dst[0] = v[j][0]; dst[1] = v[j][1]; dst[2] = v[j][2]; dst[3] = v[j][3];
        }
    }
    __syncthreads();
    if (tid < RS_ROWS * 16) {
        const int rr = tid >> 4, c = tid & 15;
        if (c < nch) {
            const float* src = xs + rr * RS_ROW + c * RS_CH;
            float p = 0.0f;
#pragma unroll 16
            for (int k = 0; k < 64; ++k) { const float q = src[k] * src[k]; p = p + q; }
            part[rr][c] = p;
        }
    }
    __syncthreads();
    if (tid < RS_ROWS) {
        float t = part[tid][0];
        for (int k = 1; k < nch; ++k) t = t + part[tid][k];
        mv[tid] = sqrtf(t / (float)d + eps);
    }
    __syncthreads();
    const float mval = mv[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = lane + 64 * j;
        if (cc < per_row && live) {
            f32x4 o;
            o[0] = v[j][0] / mval * cz_bits_f32(wb[j].x << 16);
            o[1] = v[j][1] / mval * cz_bits_f32(wb[j].x & 0xffff0000u);
            o[2] = v[j][2] / mval * cz_bits_f32(wb[j].y << 16);
            o[3] = v[j][3] / mval * cz_bits_f32(wb[j].y & 0xffff0000u);
            *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + cc * 4) = o;
        }
    }
}
hipError_t cz_launch_rmsnorm(hipStream_t st, const float* X, int ldx, const uint16_t* w, float* Y, int ldy, int M, int d, float eps,
                             const int* row_index) {
    if (M <= 0) return hipSuccess;
    if (d % 64 != 0 || d > 1024) return hipErrorInvalidValue;
    if (!row_index && M <= 2048) {
        hipLaunchKernelGGL(rmsnorm_small_kernel, dim3((M + RS_ROWS - 1) / RS_ROWS), dim3(256), 0, st, X, ldx, w, Y, ldy, M, d, eps);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 15) / 16), dim3(256), 0, st, X, ldx, w, Y, ldy, M, d, eps, row_index);
    return hipGetLastError();
}

// Y[i][:] = X[rows[i]][:]   (d % 4 == 0): the wanted rows of a packed batch, made contiguous for the last layer's tail
__global__ void gather_rows_kernel(const float* __restrict__ X, int ldx, const int* __restrict__ rows, float* __restrict__ Y, int ldy, int M, int d) {
    const int per_row = d / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / per_row), c = (int)(i % per_row);
        *reinterpret_cast<f32x4*>(Y + (size_t)m * ldy + c * 4) = *reinterpret_cast<const f32x4*>(X + (size_t)rows[m] * ldx + c * 4);
    }
}
hipError_t cz_launch_gather_rows(hipStream_t st, const float* X, int ldx, const int* rows, float* Y, int ldy, int M, int d) {
    if (M <= 0) return hipSuccess;
    if (d % 4) return hipErrorInvalidValue;
    long n = (long)M * (d / 4);
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, st, X, ldx, rows, Y, ldy, M, d);
    return hipGetLastError();
}

// RoPE (candle_nn::rotary_emb::rope, non-interleaved halves; products rounded separately, no fma) applied in place to
// the q heads of the fused QKV buffer; the k heads are rotated and written, together with v, to the per-sequence
// destination (the KV cache for lock-step decode, or back in place for segment-parallel encode).
// QKV row layout: [q: n_heads*64 | k: n_kv*64 | v: n_kv*64].
__global__ void rope_kernel(float* __restrict__ QKV, int ld, const cz_rope_desc* __restrict__ descs, int n_heads, int n_kv,
                            const float* __restrict__ cs, const float* __restrict__ sn, size_t kv_off,
                            const int* __restrict__ pos_dev) {
    cz_rope_desc d = descs[blockIdx.y];
    if (pos_dev) d.pos0 = *pos_dev;
    d.k_dst += kv_off; d.v_dst += kv_off;
    const int per_row = (n_heads + 2 * n_kv) * 32;         // one thread per rotation pair / per 2 v elements
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)d.nrows * per_row; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_row), e = (int)(i % per_row);
        const int h = e >> 5, t = e & 31;
        const int pos = d.pos0 + r;
        float* row = QKV + (size_t)(d.row0 + r) * ld;
        if (h < n_heads + n_kv) {
            float* x = row + h * 64;
            const float c = cs[(size_t)pos * 32 + t], s = sn[(size_t)pos * 32 + t];
            const float x1 = x[t], x2 = x[t + 32];
            const float a = x1 * c, b = x2 * s, cc = x1 * s, dd = x2 * c;
            const float o1 = a - b, o2 = cc + dd;
            if (h < n_heads) { x[t] = o1; x[t + 32] = o2; }
            else {
                if (d.kt_stride) {
                    float* kd = d.k_dst + (size_t)((h - n_heads) * 64) * d.kt_stride + pos;
                    kd[(size_t)t * d.kt_stride] = o1; kd[(size_t)(t + 32) * d.kt_stride] = o2;
                } else {
                    float* kd = d.k_dst + (size_t)pos * d.dst_stride + (h - n_heads) * 64;
                    kd[t] = o1; kd[t + 32] = o2;
                }
            }
        } else {
            const int hv = h - n_heads - n_kv;
            const float* v = row + (n_heads + n_kv + hv) * 64;
            float* vd = d.v_dst + (size_t)pos * d.dst_stride + hv * 64;
            if (vd != v) { vd[t] = v[t]; vd[t + 32] = v[t + 32]; }
        }
    }
}
hipError_t cz_launch_rope(hipStream_t st, float* QKV, int ld, const cz_rope_desc* descs, int nseq, int max_rows, int n_heads,
                          int n_kv, const float* cs, const float* sn, size_t kv_off, const int* pos_dev) {
    if (nseq <= 0 || max_rows <= 0) return hipSuccess;
    long n = (long)max_rows * (n_heads + 2 * n_kv) * 32;
    int bx = (int)((n + 255) / 256); if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(rope_kernel, dim3(bx, nseq), dim3(256), 0, st, QKV, ld, descs, n_heads, n_kv, cs, sn, kv_off, pos_dev);
    return hipGetLastError();
}

// SwiGLU: H[m][j] = silu(G[m][j]) * U[m][j]  with the fused gate|up GEMM output GU[m] = [G (f) | U (f)]
// (llama.rs MLP: silu(c_fc1(x)) * c_fc2(x); candle silu = x / (1 + exp(-x)))
__global__ void swiglu_kernel(const float* __restrict__ GU, float* __restrict__ H, int M, int f) {
    const int per_row = f / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / per_row), c = (int)(i % per_row);
        const f32x4 g = *reinterpret_cast<const f32x4*>(GU + (size_t)m * 2 * f + c * 4);
        const f32x4 u = *reinterpret_cast<const f32x4*>(GU + (size_t)m * 2 * f + f + c * 4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = cz_siluf(g[k]) * u[k];
        *reinterpret_cast<f32x4*>(H + (size_t)m * f + c * 4) = o;
    }
}
hipError_t cz_launch_swiglu(hipStream_t st, const float* GU, float* H, int M, int f) {
    if (M <= 0) return hipSuccess;
    long n = (long)M * (f / 4);
    int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(swiglu_kernel, dim3(blocks), dim3(256), 0, st, GU, H, M, f);
    return hipGetLastError();
}

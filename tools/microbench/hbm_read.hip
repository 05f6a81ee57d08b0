// What a read-only stream reaches on this chip, in the access shapes of dec_attn_kernel (the self-test line's largest kernel: 0.69 of the 8 TB/s peak): the ceiling its
// roofline fraction can be held against.  (a) 16-byte loads, fully coalesced, grid-stride; (b) 4-byte loads, 256 B per wave instruction, sixteen rows 8 KB apart in flight per
// wave (the transposed-K rounds); (c) 4-byte loads of 256-B rows 768 B apart (the V rounds of one KV head); each with plain and non-temporal loads, 8 / 16 waves per CU.
// Build: make -C candlezip_amd/csrc hbm_read ; run: candlezip_amd/lib/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(512) void read16_kernel(const f32x4* __restrict__ p, size_t n16, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += v;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
// block = one "stream x KV head": keys 0..nk-1; K^T rows [64 d][ctx] floats; a wave takes 64 consecutive keys and loads its 64 d-values as 4 rounds of 16 dwords
template <bool NT>
__global__ __launch_bounds__(512) void read_kt_kernel(const float* __restrict__ base, int ctx, int nk, size_t per_block_floats, float* out) {
    const float* kb = base + (size_t)blockIdx.x * per_block_floats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int k0 = wave * 64; k0 < nk; k0 += 8 * 64) {
        const int j = k0 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float r[16];
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) r[dd] = NT ? __builtin_nontemporal_load(kb + (size_t)(g * 16 + dd) * ctx + j) : kb[(size_t)(g * 16 + dd) * ctx + j];
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) acc += r[dd];
        }
    }
    if (acc == 12345.678f) out[0] = 1.0f;
}
// the same with the block count per CU pinned by dynamic LDS (dec_attn_kernel runs two 8-wave blocks per CU) and DEPTH dwords in flight per lane
template <int DEPTH>
__global__ __launch_bounds__(512) void read_kt_occ_kernel(const float* __restrict__ base, int ctx, int nk, size_t per_block_floats, float* out) {
    extern __shared__ float pad_[];
    const float* kb = base + (size_t)blockIdx.x * per_block_floats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int k0 = wave * 64; k0 < nk; k0 += 8 * 64) {
        const int j = k0 + lane;
#pragma unroll
        for (int g = 0; g < 64 / DEPTH; ++g) {
            float r[DEPTH];
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) r[dd] = __builtin_nontemporal_load(kb + (size_t)(g * DEPTH + dd) * ctx + j);
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) acc += r[dd];
        }
    }
    if (acc == 12345.678f) { out[0] = 1.0f; pad_[threadIdx.x] = acc; }
}
template <typename F>
static void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("{\"case\": \"%s\", \"ms\": %.3f, \"tb_per_s\": %.2f}\n", name, ms, bytes / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = (size_t)4 << 30;                                   // 4 GiB: past the 256 MB infinity cache
    float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    float* out; CK(hipMalloc(&out, 64));
    const size_t n16 = bytes / 16;
    for (int bpc : {2, 4, 8})
        for (int nt = 0; nt < 2; ++nt) {
            char nm[128]; snprintf(nm, sizeof nm, "16-byte loads, %d blocks of 512 per CU%s", bpc, nt ? ", non-temporal" : "");
            if (nt) run(nm, [&] { read16_kernel<true><<<256 * bpc, 512>>>((const f32x4*)buf, n16, out); }, (double)bytes);
            else run(nm, [&] { read16_kernel<false><<<256 * bpc, 512>>>((const f32x4*)buf, n16, out); }, (double)bytes);
        }
    // dec_attn's K phase: 1536 blocks (512 streams x 3 KV heads), ctx 2048, 768 keys: 64 rows x 768 keys x 4 B = 196 608 B per block, rows 8 KB apart
    const int ctx = 2048, nk = 768;
    const size_t per_block = (size_t)64 * ctx;                              // floats a block's K^T rows span (512 KB)
    const int nblk = (int)(bytes / (per_block * sizeof(float)));            // 8 192 blocks: the whole buffer, 1.6 GB read (past the caches)
    if ((size_t)nblk * per_block * sizeof(float) > bytes || nk > ctx) { fprintf(stderr, "shape check failed\n"); return 1; }
    for (int nt = 0; nt < 2; ++nt) {
        char nm[128]; snprintf(nm, sizeof nm, "transposed-K rounds (16 dwords in flight per lane, rows 8 KB apart), 768 of 2048 keys%s", nt ? ", non-temporal" : "");
        if (nt) run(nm, [&] { read_kt_kernel<true><<<nblk, 512>>>(buf, ctx, nk, per_block, out); }, (double)nblk * 64 * nk * 4);
        else run(nm, [&] { read_kt_kernel<false><<<nblk, 512>>>(buf, ctx, nk, per_block, out); }, (double)nblk * 64 * nk * 4);
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_kt_occ_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_kt_occ_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(read_kt_occ_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int bpc : {1, 2, 3, 4}) {
        const size_t lds = (size_t)(160 / bpc - 2) * 1024;                  // bpc blocks per CU by LDS
        char nm[160];
        snprintf(nm, sizeof nm, "transposed-K rounds, non-temporal, %d blocks of 8 waves per CU, 16 dwords in flight per lane", bpc);
        run(nm, [&] { read_kt_occ_kernel<16><<<nblk, 512, lds>>>(buf, ctx, nk, per_block, out); }, (double)nblk * 64 * nk * 4);
        snprintf(nm, sizeof nm, "transposed-K rounds, non-temporal, %d blocks of 8 waves per CU, 32 dwords in flight per lane", bpc);
        run(nm, [&] { read_kt_occ_kernel<32><<<nblk, 512, lds>>>(buf, ctx, nk, per_block, out); }, (double)nblk * 64 * nk * 4);
        snprintf(nm, sizeof nm, "transposed-K rounds, non-temporal, %d blocks of 8 waves per CU, 64 dwords in flight per lane", bpc);
        run(nm, [&] { read_kt_occ_kernel<64><<<nblk, 512, lds>>>(buf, ctx, nk, per_block, out); }, (double)nblk * 64 * nk * 4);
    }
    return 0;
}

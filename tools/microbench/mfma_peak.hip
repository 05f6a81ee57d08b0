// Measures what the f32 matrix pipe of gfx950 delivers with nothing else in the way: waves that only issue v_mfma_f32_32x32x2_f32 (or 16x16x4) on NACC
// independent accumulators, for 1..4 waves per SIMD, with and without ds_read_b128 traffic next to them.  SURVEY.md §8(d) prices the encode's GEMMs
// against 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz): this is the "confirm with a microbench on the box" that line asks for.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o candlezip_amd/lib/mfma_peak ; run: candlezip_amd/lib/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ unsigned long long g_clk[3];

template <int NACC, int LDSR>
__global__ __launch_bounds__(256) void mfma32_kernel(float* out, int iters, float a0, float b0) {
    __shared__ float lds[LDSR ? 8192 : 64];
    for (int i = threadIdx.x; i < (LDSR ? 8192 : 64); i += blockDim.x) lds[i] = a0 * i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDSR) {                                                           // one 16-byte LDS read per lane per LDSR MFMAs, used as the next operands
            const f32x4 v = *(const f32x4*)&lds[((threadIdx.x * 4 + it * 256) & 8188)];
            a += v[0]; b += v[1];
        }
#pragma unroll
        for (int r = 0; r < (LDSR ? LDSR / NACC : 1); ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&g_clk[0], c1 - c0); atomicAdd(&g_clk[1], r1 - r0); atomicAdd(&g_clk[2], 1ull); }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// waves 0..3 of a 512-thread block stream MFMAs, waves 4..7 (one per SIMD beside them) run a dependent-free VALU loop: how many vector issues does a wave
// get in the shadow of another wave's MFMA stream?  out[] gets the VALU waves' cycles per VALU instruction.
template <int MODE>   // 0: VALU waves alone (MFMA waves exit), 1: beside 32x32x2 MFMAs, 2: beside 16x16x4 MFMAs
__global__ __launch_bounds__(512) void valu_shadow_kernel(float* out, int iters, float a0) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (MODE == 0) return;
        f32x16 acc[4]; f32x4 acc4[4];
        for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) acc[j][e] = 0.f; for (int e = 0; e < 4; ++e) acc4[j][e] = 0.f; }
        float a = a0 + threadIdx.x, b = a0;
        const int n = MODE == 1 ? iters : iters * 2;
        for (int it = 0; it < n; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                else acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[j], 0, 0, 0);
            }
        float s = 0.f;
        for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) s += acc[j][e]; for (int e = 0; e < 4; ++e) s += acc4[j][e]; }
        if (s == 12345.678f) out[threadIdx.x] = s;
        return;
    }
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = a0 * (j + 1) + threadIdx.x;
    const int n = iters / 4;                                                 // the VALU waves finish well inside the MFMA waves' run (iters * 4 MFMAs * 64 cycles)
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 1.0000001f, a0);      // 8 independent chains, 32 v_fma per iteration
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&g_clk[0], c1 - c0); atomicAdd(&g_clk[2], 1ull); }
}

template <int MODE>
static void run_shadow(const char* name, int iters) {
    float* out; CK(hipMalloc(&out, 4096));
    unsigned long long z[3] = {0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z));
    valu_shadow_kernel<MODE><<<256, 512>>>(out, iters, 1e-9f);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_clk), sizeof z));
    printf("{\"case\": \"%s\", \"cycles_per_valu_instruction\": %.2f}\n", name, (double)z[0] / z[2] / ((double)(iters / 4) * 32));
    CK(hipFree(out));
}

// every wave issues NV independent v_fma_f32 after each MFMA (its own VALU work between its own MFMAs): the MFMA rate that is left, for W waves per SIMD
template <int NV, int SHAPE>   // SHAPE 32: 32x32x2 (64 cycles), 16: 16x16x4 (32 cycles)
__global__ __launch_bounds__(256) void mix_kernel(float* out, int iters, float a0) {
    f32x16 acc[4]; f32x4 acc4[4];
    for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) acc[j][e] = 0.f; for (int e = 0; e < 4; ++e) acc4[j][e] = 0.f; }
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = a0 * (j + 1) + threadIdx.x;
    float a = a0 + threadIdx.x, b = a0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (SHAPE == 32) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            else acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[j], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) { x[(j * NV + v) & 7] = __builtin_fmaf(x[(j * NV + v) & 7], 1.0000001f, a0); asm volatile("" : "+v"(x[(j * NV + v) & 7])); }
        }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) s += acc[j][e]; for (int e = 0; e < 4; ++e) s += acc4[j][e]; }
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NV, int SHAPE>
static void run_mix(int waves_per_simd, int iters) {
    float* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mix_kernel<NV, SHAPE><<<256 * waves_per_simd, 256>>>(out, iters / 8, 1e-9f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    mix_kernel<NV, SHAPE><<<256 * waves_per_simd, 256>>>(out, iters, 1e-9f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * waves_per_simd * 4 * iters * 4 * (SHAPE == 32 ? 4096.0 : 2048.0);
    printf("{\"case\": \"mfma %s + %d v_fma per mfma in the same wave\", \"waves_per_simd\": %d, \"tflops\": %.1f}\n", SHAPE == 32 ? "32x32x2" : "16x16x4", NV, waves_per_simd, flops / (ms * 1e-3) / 1e12);
    CK(hipFree(out));
}

template <typename F>
static void run(const char* name, F launch, int blocks_per_cu, int threads, double flops_per_wave_iter, int mf_per_iter, int iters) {
    float* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    launch(grid, threads, out, iters / 8);                                    // warm
    CK(hipDeviceSynchronize());
    unsigned long long z[3] = {0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) launch(grid, threads, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_clk), sizeof z));
    const double waves = (double)grid * threads / 64, tf = waves * iters * flops_per_wave_iter / (ms * 1e-3) / 1e12;
    const double ghz = z[1] ? (double)z[0] / (z[1] * 10.0) : 0.0;                      // s_memrealtime: 100 MHz
    const double cyc_per_mfma = z[2] ? (double)z[0] / z[2] / ((double)iters * mf_per_iter) * 1.0 : 0.0;   // wall cycles of ONE wave per MFMA it issued (x waves per SIMD = pipe cycles)
    printf("{\"case\": \"%s\", \"waves_per_simd\": %.2f, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f, \"wave_cycles_per_mfma\": %.1f}\n", name, blocks_per_cu * threads / 256.0, ms, tf, ghz, cyc_per_mfma);
    CK(hipFree(out));
}

int main() {
    const int it = 20000;
#define L32(NACC, LDSR) [](int g, int t, float* o, int n) { mfma32_kernel<NACC, LDSR><<<g, t>>>(o, n, 1e-9f, 1e-9f); }
#define L16(NACC) [](int g, int t, float* o, int n) { mfma16_kernel<NACC><<<g, t>>>(o, n, 1e-9f, 1e-9f); }
    run("32x32x2 acc4 1 wave/simd", L32(4, 0), 1, 256, 4 * 4096.0, 4, it);
    run("32x32x2 acc4 2 waves/simd", L32(4, 0), 2, 256, 4 * 4096.0, 4, it);
    run("32x32x2 acc4 3 waves/simd", L32(4, 0), 3, 256, 4 * 4096.0, 4, it);
    run("32x32x2 acc4 4 waves/simd", L32(4, 0), 4, 256, 4 * 4096.0, 4, it);
    run("32x32x2 acc1 1 wave/simd", L32(1, 0), 1, 256, 4096.0, 1, it * 4);
    run("32x32x2 acc1 2 waves/simd", L32(1, 0), 2, 256, 4096.0, 1, it * 4);
    run("32x32x2 acc2 2 waves/simd", L32(2, 0), 2, 256, 2 * 4096.0, 2, it * 2);
    run("32x32x2 acc4 2 waves/simd + ds_read_b128 per 4 mfma", L32(4, 4), 2, 256, 4 * 4096.0, 4, it);
    run("32x32x2 acc4 2 waves/simd + ds_read_b128 per 8 mfma", L32(4, 8), 2, 256, 8 * 4096.0, 8, it / 2);
    run("16x16x4 acc4 2 waves/simd", L16(4), 2, 256, 4 * 2048.0, 4, it * 2);
    run("16x16x4 acc8 2 waves/simd", L16(8), 2, 256, 8 * 2048.0, 8, it);
    run("16x16x4 acc4 4 waves/simd", L16(4), 4, 256, 4 * 2048.0, 4, it * 2);
    run_mix<1, 32>(1, 20000); run_mix<2, 32>(1, 20000); run_mix<4, 32>(1, 20000); run_mix<8, 32>(1, 20000); run_mix<12, 32>(1, 20000); run_mix<16, 32>(1, 20000);
    run_mix<1, 32>(2, 20000); run_mix<2, 32>(2, 20000); run_mix<4, 32>(2, 20000); run_mix<8, 32>(2, 20000); run_mix<12, 32>(2, 20000);
    run_mix<2, 32>(4, 20000); run_mix<4, 32>(4, 20000); run_mix<8, 32>(4, 20000);
    run_mix<1, 16>(1, 40000); run_mix<2, 16>(1, 40000); run_mix<4, 16>(1, 40000); run_mix<8, 16>(1, 40000);
    run_mix<1, 16>(2, 40000); run_mix<2, 16>(2, 40000); run_mix<4, 16>(2, 40000); run_mix<8, 16>(2, 40000);
    run_mix<1, 16>(4, 40000); run_mix<2, 16>(4, 40000); run_mix<4, 16>(4, 40000);
    run_shadow<0>("v_fma_f32 loop, one wave per SIMD, nothing beside it", 40000);
    run_shadow<1>("v_fma_f32 loop beside a wave streaming 32x32x2 f32 MFMAs on the same SIMD", 40000);
    run_shadow<2>("v_fma_f32 loop beside a wave streaming 16x16x4 f32 MFMAs on the same SIMD", 40000);
    return 0;
}

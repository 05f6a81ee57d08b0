// kernels_probe.hip — per-kernel entry points for the GPU parity tests (host pointers in, host pointers out).
// Not part of the drop-in boundary; they exist so that tests/ can compare every kernel with the oracle in isolation
// (a whole-model tolerance would hide an O(1)-wrong sub-stage).
#include <string.h>
#include <vector>
#include "cz_host.h"
#include "cz_kernels.h"
#include "cz_cdf_common.h"

#define PCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cz_set_error("HIP error %s at %s:%d", hipGetErrorString(_e), __FILE__, __LINE__); goto fail; } } while (0)

struct DevMem {
    std::vector<void*> ptrs;
    void* alloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n ? n : 4) != hipSuccess) return nullptr; ptrs.push_back(p); return p; }
    ~DevMem() { for (auto p : ptrs) { hipError_t e = hipFree(p); (void)e; } }
};

__global__ void scalar_fn_kernel(int fn, const float* x, const float* y, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y ? y[i] : 0.0f, r;
    switch (fn) {
        case 0: r = cz_expf(a); break;
        case 1: r = cz_logf(a); break;
        case 2: r = cz_tanhf(a); break;
        case 3: r = a / b; break;
        case 4: r = sqrtf(a); break;
        case 5: r = cz_siluf(a); break;
        case 6: r = cz_sigmoidf(a); break;
        case 7: r = a * b; break;
        case 8: r = cz_fma(a, b, a); break;
        default: r = 0.0f;
    }
    out[i] = r;
}
__global__ void f64_fn_kernel(int fn, const double* x, const double* y, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], b = y[i], r;
    switch (fn) {
        case 0: r = a / b; break;
        case 1: r = a + b; break;
        case 2: r = floor(a * 1073741824.0); break;
        case 3: r = a * b; break;
        case 4: r = div_by(a, b, 1.0 / b); break;          // the CDF kernels' division by a row-wide divisor (cz_cdf_common.h)
        default: r = 0.0;
    }
    out[i] = r;
}

// every float bit pattern through the device form of cz_expf and through its definition: out[0] = number of inputs whose results differ in any bit,
// out[1] = the first such input's bits (lowest), out[2] = inputs compared
__global__ void expf_exhaustive_kernel(unsigned long long* out) {
    unsigned long long bad = 0, first = ~0ull;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = cz_bits_f32((uint32_t)u);
        if (cz_f32_bits(cz_expf(x)) != cz_f32_bits(cz_expf_ref(x))) { ++bad; if (u < first) first = u; }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
    if (threadIdx.x == 0) atomicAdd(&out[2], 0ull);
}
extern "C" cz_status cz_debug_expf_exhaustive(unsigned long long* out3) {
    DevMem m;
    unsigned long long* d = (unsigned long long*)m.alloc(24);
    if (!d) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    const unsigned long long init[3] = {0, ~0ull, 1ull << 32};
    PCHK(hipMemcpy(d, init, 24, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(expf_exhaustive_kernel, dim3(4096), dim3(256), 0, 0, d);
    PCHK(hipGetLastError());
    PCHK(hipMemcpy(out3, d, 24, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}
extern "C" cz_status cz_debug_scalar(int fn, const float* x, const float* y, float* out, int n) {
    DevMem m;
    float *dx = (float*)m.alloc(n * 4), *dy = (float*)m.alloc(n * 4), *dout = (float*)m.alloc(n * 4);
    if (!dx || !dy || !dout) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    if (y) PCHK(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scalar_fn_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, y ? dy : nullptr, dout, n);
    PCHK(hipGetLastError());
    PCHK(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}
extern "C" cz_status cz_debug_f64(int fn, const double* x, const double* y, double* out, int n) {
    DevMem m;
    double *dx = (double*)m.alloc(n * 8), *dy = (double*)m.alloc(n * 8), *dout = (double*)m.alloc(n * 8);
    if (!dx || !dy || !dout) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(dy, y, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(f64_fn_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, dy, dout, n);
    PCHK(hipGetLastError());
    PCHK(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// C = A . W^T (+ R), A f32 [M][K], W bf16 [N][K]
extern "C" cz_status cz_debug_gemm(const float* A, const uint16_t* W, const float* R, float* C, int M, int N, int K, int force_big) {
    DevMem m;
    float *dA = (float*)m.alloc((size_t)M * K * 4), *dC = (float*)m.alloc((size_t)M * N * 4), *dR = R ? (float*)m.alloc((size_t)M * N * 4) : nullptr;
    uint16_t* dW = (uint16_t*)m.alloc((size_t)N * K * 2);
    if (!dA || !dC || !dW || (R && !dR)) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(dW, W, (size_t)N * K * 2, hipMemcpyHostToDevice));
    if (R) PCHK(hipMemcpy(dR, R, (size_t)M * N * 4, hipMemcpyHostToDevice));
    PCHK(hipMemset(dC, 0xFF, (size_t)M * N * 4));
    PCHK(cz_launch_gemm(0, dA, K, dW, dC, N, dR, N, M, N, K, force_big));
    PCHK(hipDeviceSynchronize());
    PCHK(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

extern "C" cz_status cz_debug_rmsnorm(const float* X, const uint16_t* w, float* Y, int M, int d, float eps) {
    DevMem m;
    float *dX = (float*)m.alloc((size_t)M * d * 4), *dY = (float*)m.alloc((size_t)M * d * 4);
    uint16_t* dw = (uint16_t*)m.alloc((size_t)d * 2);
    if (!dX || !dY || !dw) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dX, X, (size_t)M * d * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(dw, w, (size_t)d * 2, hipMemcpyHostToDevice));
    PCHK(cz_launch_rmsnorm(0, dX, d, dw, dY, d, M, d, eps, nullptr));
    PCHK(hipDeviceSynchronize());
    PCHK(hipMemcpy(Y, dY, (size_t)M * d * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// logits [rows][V] -> (c_lo, c_hi) of sym[rows]
extern "C" cz_status cz_debug_cdf_pairs(const float* logits, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi, int rows, int V, int mode, int serial,
                                        unsigned* fallbacks) {
    DevMem m;
    float* dl = (float*)m.alloc((size_t)rows * V * 4);
    uint32_t *ds = (uint32_t*)m.alloc(rows * 4), *dlo = (uint32_t*)m.alloc(rows * 4), *dhi = (uint32_t*)m.alloc(rows * 4);
    double* dscr = (!serial && cz_cdf_par_supported(V, mode)) ? (double*)m.alloc((size_t)(rows < CZ_CDF_PAR_MAX_GRID ? rows : CZ_CDF_PAR_MAX_GRID) * (V + 256) * 16) : nullptr;
    if (!dl || !ds || !dlo || !dhi || (!serial && cz_cdf_par_supported(V, mode) && !dscr)) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dl, logits, (size_t)rows * V * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(ds, sym, rows * 4, hipMemcpyHostToDevice));
    {
        unsigned* dfb = (unsigned*)m.alloc(64);
        if (!dfb) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
        PCHK(hipMemset(dfb, 0, 64));
        if (serial) PCHK(cz_launch_cdf_pairs_serial(0, dl, V, ds, dlo, dhi, rows, V, mode));
        else PCHK(cz_launch_cdf_pairs(0, dl, V, ds, dlo, dhi, rows, V, mode, dfb, dscr));
        PCHK(hipDeviceSynchronize());
        if (fallbacks) PCHK(hipMemcpy(fallbacks, dfb, 4, hipMemcpyDeviceToHost));
    }
    PCHK(hipMemcpy(c_lo, dlo, rows * 4, hipMemcpyDeviceToHost));
    PCHK(hipMemcpy(c_hi, dhi, rows * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// decode `steps` symbols of ONE stream from `payload`, using logits[step][V] (teacher-provided), on the device
extern "C" cz_status cz_debug_decode_symbols(const float* logits, int steps, int V, int mode, const uint8_t* payload, size_t len,
                                             uint32_t* out_syms, int serial) {
    DevMem m;
    float* dl = (float*)m.alloc((size_t)steps * V * 4);
    uint8_t* dp = (uint8_t*)m.alloc(len + 16);
    uint32_t* dout = (uint32_t*)m.alloc((size_t)(steps + 1) * 4);
    uint32_t* dsym = (uint32_t*)m.alloc(64);
    cz_dec_state* dst = (cz_dec_state*)m.alloc(sizeof(cz_dec_state));
    cz_dec_state hs;
    double* dscr = (!serial && cz_cdf_par_supported(V, mode)) ? (double*)m.alloc((size_t)(V + 256) * 16) : nullptr;
    if (!dl || !dp || !dout || !dsym || !dst || (!serial && cz_cdf_par_supported(V, mode) && !dscr)) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dl, logits, (size_t)steps * V * 4, hipMemcpyHostToDevice));
    if (len) PCHK(hipMemcpy(dp, payload, len, hipMemcpyHostToDevice));
    memset(&hs, 0, sizeof hs);
    hs.in = dp; hs.in_len = len; hs.out_ids = dout; hs.token_count = (unsigned long long)steps; hs.active = 1; hs.final_input = 1;
    PCHK(hipMemcpy(dst, &hs, sizeof hs, hipMemcpyHostToDevice));
    PCHK(cz_launch_dec_init(0, dst, 1));
    for (int i = 0; i < steps; ++i) {
        if (serial) PCHK(cz_launch_decode_symbol_serial(0, dl + (size_t)i * V, V, dst, dsym, 1, V, mode, (unsigned long long)i, nullptr));
        else PCHK(cz_launch_decode_symbol(0, dl + (size_t)i * V, V, dst, dsym, 1, V, mode, (unsigned long long)i, nullptr, nullptr, dscr));
    }
    PCHK(hipDeviceSynchronize());
    PCHK(hipMemcpy(out_syms, dout + 1, (size_t)steps * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// rows at positions pos0.. attend to K/V rows [0, pos0+m]; Q/O [M][nh*64], K/V [pos0+M][nkv*64]
extern int g_attn_v1;
// test hook: which prefill-attention kernel cz_launch_attention picks (0 = by shape, 1 = v1, 2 = v3)
extern "C" void cz_debug_attn_variant(int v) { g_attn_v1 = v; }
extern "C" cz_status cz_debug_attention(const float* Q, const float* K, const float* V, float* O, int M, int pos0, int nh, int nkv) {
    DevMem m;
    const int nk = pos0 + M;
    float *dQ = (float*)m.alloc((size_t)M * nh * 64 * 4), *dO = (float*)m.alloc((size_t)M * nh * 64 * 4);
    float *dK = (float*)m.alloc((size_t)nk * nkv * 64 * 4), *dV = (float*)m.alloc((size_t)nk * nkv * 64 * 4);
    cz_seq_desc* dd = (cz_seq_desc*)m.alloc(sizeof(cz_seq_desc));
    cz_seq_desc hd;
    if (!dQ || !dO || !dK || !dV || !dd) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    PCHK(hipMemcpy(dQ, Q, (size_t)M * nh * 64 * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(dK, K, (size_t)nk * nkv * 64 * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(dV, V, (size_t)nk * nkv * 64 * 4, hipMemcpyHostToDevice));
    PCHK(hipMemset(dO, 0xFF, (size_t)M * nh * 64 * 4));
    hd.row0 = 0; hd.nrows = M; hd.pos0 = pos0; hd.kv_stride = nkv * 64; hd.k = dK; hd.v = dV; hd.kt_stride = 0; hd.pad_ = 0;
    PCHK(hipMemcpy(dd, &hd, sizeof hd, hipMemcpyHostToDevice));
    PCHK(cz_attention_prepare(2048));
    PCHK(cz_launch_attention(0, dQ, nh * 64, dO, nh * 64, dd, 1, M, nk, nh, nkv, 0, nullptr));
    PCHK(hipDeviceSynchronize());
    PCHK(hipMemcpy(O, dO, (size_t)M * nh * 64 * 4, hipMemcpyDeviceToHost));
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// ---- measurement aid (tools/probe_chain.py): the cost of one node in a chain of dependent kernels, as the decode step of a single stream is one.
// kind 0: one lane adds 1 to a word (nothing but the dependency); kind 1: `blocks` blocks of 256 threads each stream `kb` KB of a buffer and one
// lane per block adds its sum to a word (a weight-streaming GEMV's memory shape without its arithmetic).
__global__ void probe_chain_kernel(float* acc, const float* __restrict__ buf, int kb) {
    float s = 0.0f;
    if (kb > 0) {
        const float4* p = reinterpret_cast<const float4*>(buf) + (size_t)blockIdx.x * (kb * 64) + threadIdx.x;
        for (int i = 0; i < kb * 64 / 256; ++i) { const float4 v = p[i * 256]; s += v.x + v.y + v.z + v.w; }
    }
    if (threadIdx.x == 0) acc[blockIdx.x] = acc[blockIdx.x] + 1.0f + s * 0.0f;
}
extern "C" cz_status cz_debug_probe_chain(int nodes, int blocks, int kb, int iters, int use_graph, double* us_per_node) {
    DevMem m;
    float* acc = (float*)m.alloc((size_t)(blocks > 0 ? blocks : 1) * 4);
    const size_t bytes = (size_t)(blocks > 0 ? blocks : 1) * (kb > 0 ? kb : 1) * 1024;
    float* buf = (float*)m.alloc(bytes);
    if (!acc || !buf || nodes <= 0 || iters <= 0 || blocks <= 0 || (kb && kb % 4)) { cz_set_error("cz_debug_probe_chain: bad argument"); return CZ_ERR_ARG; }
    hipStream_t st = nullptr; hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.0f;
    PCHK(hipMemset(acc, 0, (size_t)blocks * 4)); PCHK(hipMemset(buf, 0, bytes));
    PCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    PCHK(hipEventCreate(&e0)); PCHK(hipEventCreate(&e1));
    if (use_graph) {
        PCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int n = 0; n < nodes; ++n) hipLaunchKernelGGL(probe_chain_kernel, dim3(blocks), dim3(256), 0, st, acc, buf, kb);
        PCHK(hipStreamEndCapture(st, &g));
        PCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        PCHK(hipGraphLaunch(ge, st)); PCHK(hipStreamSynchronize(st));
    }
    PCHK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) {
        if (use_graph) PCHK(hipGraphLaunch(ge, st));
        else for (int n = 0; n < nodes; ++n) hipLaunchKernelGGL(probe_chain_kernel, dim3(blocks), dim3(256), 0, st, acc, buf, kb);
    }
    PCHK(hipEventRecord(e1, st));
    PCHK(hipStreamSynchronize(st));
    PCHK(hipEventElapsedTime(&ms, e0, e1));
    *us_per_node = 1e3 * ms / ((double)iters * nodes);
    if (ge) hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

// ---- measurement aid (tools/probe_persist.py): what ONE all-to-all hand-off costs INSIDE a launch.  G resident blocks pass through `stages` stages;
// in each a block (1) requests `kb` KB of "weights" (independent of the hand-off, so before it), (2) waits until every block has finished the previous
// stage (one monotonic counter, agent scope), (3) reads the whole `xn`-float activation vector the others wrote, (4) writes its own slice of the next
// one, (5) fences and counts itself in.  `stride` = 8 keeps only every eighth block (the others return at once): workgroups are dealt round-robin to
// the eight XCDs, so the survivors share one L2.  Every wait is bounded (err = 1 and the block leaves), so the grid always drains.
__global__ __launch_bounds__(256) void probe_persist_kernel(float* x, unsigned* ctr, const float4* __restrict__ w, int stages, int stride, int kb, int xn, int wslabs, int* err) {
    if (blockIdx.x % stride) return;
    const int g = blockIdx.x / stride, G = gridDim.x / stride, tid = threadIdx.x;
    const int per = (xn + G - 1) / G, n4 = kb * 64 / 256;           // outputs per block; float4 of weights per thread
    __shared__ float red[256];
    __shared__ int bail;
    if (tid == 0) bail = 0;
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        float4 wv[16];
        const float4* wp = w + ((size_t)(s % wslabs) * G + g) * (size_t)(kb * 64) + tid;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < n4) wv[i] = wp[i * 256];
        if (tid == 0 && s > 0) {
            const unsigned target = (unsigned)s * (unsigned)G;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { bail = 1; *err = 1; break; }
            }
        }
        __syncthreads();
        if (bail) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* xin = x + (size_t)(s & 1) * xn;
        float* xout = x + (size_t)((s + 1) & 1) * xn;
        float a = 0.0f;
        for (int i = tid; i < xn; i += 256) a += xin[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < n4) a += (wv[i].x + wv[i].y + wv[i].z + wv[i].w) * 1e-9f;
        red[tid] = a;
        __syncthreads();
        if (tid < 64) { float t = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192]; for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o); if (tid == 0) red[0] = t; }
        __syncthreads();
        const float tot = red[0];
        if (tid < per && g * per + tid < xn) xout[g * per + tid] = 0.25f * xin[g * per + tid] + tot * 1e-6f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
extern "C" cz_status cz_debug_probe_persist(int G, int stride, int stages, int kb, int xn, int iters, double* us_per_stage) {
    DevMem m;
    const int wslabs = 150;
    if (G <= 0 || G > 512 || (stride != 1 && stride != 8) || stages <= 0 || kb < 0 || kb > 64 || (kb % 4) || xn <= 0 || xn > 8192 || iters <= 0) { cz_set_error("cz_debug_probe_persist: bad argument"); return CZ_ERR_ARG; }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { cz_set_error("no device"); return CZ_ERR_HIP; }
    if (G * stride > 2 * cus) { cz_set_error("cz_debug_probe_persist: %d blocks (stride %d) cannot all be resident on %d CUs", G, stride, cus); return CZ_ERR_ARG; }   // (two 256-thread blocks per CU fit easily)
    const size_t wbytes = (size_t)wslabs * G * (kb ? kb : 1) * 1024;
    float* x = (float*)m.alloc((size_t)2 * xn * 4);
    unsigned* ctr = (unsigned*)m.alloc(64);
    int* err = (int*)m.alloc(64);
    float4* w = (float4*)m.alloc(wbytes);
    if (!x || !ctr || !err || !w) { cz_set_error("hipMalloc failed"); return CZ_ERR_HIP; }
    hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.0f, total = 0.0f; int herr = 0;
    PCHK(hipMemset(x, 0, (size_t)2 * xn * 4)); PCHK(hipMemset(w, 0, wbytes)); PCHK(hipMemset(err, 0, 64));
    PCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    PCHK(hipEventCreate(&e0)); PCHK(hipEventCreate(&e1));
    for (int it = 0; it <= iters; ++it) {                                  // (iteration 0 is the warm-up)
        PCHK(hipMemsetAsync(ctr, 0, 64, st));
        PCHK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(probe_persist_kernel, dim3(G * stride), dim3(256), 0, st, x, ctr, w, stages, stride, kb, xn, wslabs, err);
        PCHK(hipEventRecord(e1, st));
        PCHK(hipStreamSynchronize(st));
        PCHK(hipEventElapsedTime(&ms, e0, e1));
        if (it) total += ms;
    }
    PCHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    if (herr) { cz_set_error("cz_debug_probe_persist: a hand-off timed out (blocks not co-resident?)"); return CZ_ERR_STATE; }
    *us_per_stage = 1e3 * total / ((double)iters * stages);
    return CZ_OK;
fail:
    return CZ_ERR_HIP;
}

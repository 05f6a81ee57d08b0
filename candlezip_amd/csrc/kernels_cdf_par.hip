// kernels_cdf_par.hip — exact PARALLEL evaluation of the reference's sequential f64 chains (SmolLM mode).
//
// softmax_pdf (src/main.rs:784-801) sums exp() left to right in f64 and quantize_pdf_to_cdf (:805-824) floors a
// left-to-right running sum of the pdf; the payload depends on every rounding of those two 49 152-long chains, so
// they cannot be reordered.  This kernel evaluates them with one 512-thread workgroup per row and is bit-identical
// to the serial walk (kernels_cdf.hip), which the GPU tests check on adversarial rows.
//
// Method (executable model: tests/seqsum_model.py; DESIGN.md "Exact parallel CDF"): thread t owns the contiguous chunk
// [t*L, (t+1)*L).  While acc stays inside one binade [2^b, 2^(b+1)) every add rounds on the grid u = 2^(b-52) and a
// tie's direction depends only on the parity of acc/u, so the chunk maps acc_in -> acc_in + D[parity(acc_in)].
// D0, D1 come from running the TRUE f64 chain over the chunk from the reference starts 2^b and 2^b + u.  An ordered
// resolve (one lane per wave composes 64 transducers, then thread 0 walks the 8 waves) applies the transducers where
// acc_in is verified to sit in binade b and acc_in + D < 2^(b+1); any other chunk ("dirty": binade crossing, start at
// 0) is evaluated by its true chain from a scratch copy of its terms, kept by every chunk that is not provably clean.
// Finally every thread re-runs its true chain from its exact acc_in.  If the resolver ever meets a dirty chunk without
// a copy (the 1e-9 margin argument failed) the row falls back to the serial walk, so the result is exact regardless.
//
// Algorithmic traffic = vocab * 4 B of logits per row.  Bound: HBM for encode batches, latency for lock-step decode.
#include "cz_cdf_common.h"

#define T_PAR 1024
#define LMAX 48
#define NWAVE (T_PAR / 64)
#define POOL_SLOTS 60
#define B_INVALID (-100000)
#define B_EMPTY 100000

__device__ __forceinline__ int expo_f64(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ff) - 1023; }   // normal v > 0
__device__ __forceinline__ double pow2_f64(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }
__device__ __forceinline__ int parity_f64(double v) { return (int)(__double_as_longlong(v) & 1); }                    // LSB of the significand
__device__ __forceinline__ uint32_t quantize_u(double acc) {   // src/main.rs:812-818; the monotone clamp is implied (acc never decreases)
    double f = floor(acc * CDF_TOTAL);
    if (f != f || f < 0.0) return 0u;
    if (f > CDF_TOTAL) return CDF_TOTAL_U;
    return (uint32_t)f;
}

struct par_smem {
    double D0[T_PAR], D1[T_PAR], acc_in[T_PAR];
    int bexp[T_PAR];
    unsigned char has_copy[T_PAR];   // 0 = none, 1..POOL = LDS pool slot + 1, 255 = global scratch
    double pool[POOL_SLOTS][LMAX];   // copies of the terms of chunks that are not provably clean (usually < 30 per row)
    unsigned pool_next;
    double wsum[NWAVE], W0[NWAVE], W1[NWAVE], win[NWAVE];
    int wb[NWAVE], wfast[NWAVE];
    double total;
    float redf[NWAVE];
    int flag, L, n;
    unsigned found;
    uint32_t value, lo, hi;
    int ok;
};

__device__ __forceinline__ double block_excl_scan(double v, par_smem& sm) {     // approximate (any order): only guesses binades
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { double y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) sm.wsum[wave] = inc;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += sm.wsum[w];
    __syncthreads();
    return base + inc - v;
}

// Exact left-to-right sum of the block's terms xf(k) (thread t owns terms first..first+cnt-1, k < cnt <= LMAX).
// Returns the exact total; with want_acc_in also the exact running sum before this thread's first term.
// csum = APPROXIMATE sum of this thread's terms (any rounding; it only feeds the binade guesses).
template <typename XF>
__device__ __forceinline__ double exact_seq_sum(XF xf, double csum, int cnt, int first, par_smem& sm, double* scratch, bool want_acc_in, double* acc_in_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sm.pool_next = 0;
    const double pstart = block_excl_scan(csum, sm);            // (contains barriers: the reset above is visible afterwards)
    const double pend = pstart + csum;
    int b = B_INVALID; double d0 = 0.0, d1 = 0.0;
    bool copy = false;
    if (cnt == 0) b = B_EMPTY;
    else {
        copy = true;
        if (pstart > 1e-290 && pend < 1e290) {
            const int bs = expo_f64(pstart);
            const double r0 = pow2_f64(bs), u = pow2_f64(bs - 52), r1 = r0 + u, top = pow2_f64(bs + 1);
            double e0 = r0, e1 = r1;
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cnt) { const double x = xf(k); e0 = e0 + x; e1 = e1 + x; }
            if (e0 < top && e1 < top) {
                b = bs; d0 = e0 - r0; d1 = e1 - r1;
                // the approximate prefix is within ~n*2^-53 (< 1e-11) relative of the true one, so inside this margin the true
                // acc is certainly in binade bs across the whole chunk and the transducer will be accepted: no copy needed
                if (pstart > r0 * (1.0 + 1e-9) && pend < top * (1.0 - 1e-9)) copy = false;
            }
        }
    }
    sm.bexp[tid] = b; sm.D0[tid] = d0; sm.D1[tid] = d1;
    unsigned char where = 0;
    if (copy) {
        const unsigned slot = atomicAdd(&sm.pool_next, 1u);
        if (slot < POOL_SLOTS) {
            where = (unsigned char)(slot + 1);
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cnt) sm.pool[slot][k] = xf(k);
        } else {
            where = 255;
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cnt) scratch[first + k] = xf(k);
        }
    }
    sm.has_copy[tid] = where;
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {                                             // level 1: compose the wave's 64 transducers
        int wbv = B_EMPTY; bool ok = true;
        for (int j = 0; j < 64; ++j) {
            const int bj = sm.bexp[wave * 64 + j];
            if (bj == B_EMPTY) continue;
            if (bj == B_INVALID) { ok = false; break; }
            if (wbv == B_EMPTY) wbv = bj; else if (wbv != bj) { ok = false; break; }
        }
        double w0 = 0.0, w1 = 0.0;
        if (ok && wbv != B_EMPTY) {
            const double inv_u = pow2_f64(52 - wbv);
            int p0 = 0, p1 = 1;
            for (int j = 0; j < 64; ++j) {
                const int t = wave * 64 + j;
                if (sm.bexp[t] == B_EMPTY) continue;
                const double a = p0 ? sm.D1[t] : sm.D0[t]; w0 += a; p0 ^= (int)((long long)(a * inv_u) & 1);
                const double c = p1 ? sm.D1[t] : sm.D0[t]; w1 += c; p1 ^= (int)((long long)(c * inv_u) & 1);
            }
        }
        sm.wb[wave] = ok ? wbv : B_INVALID; sm.W0[wave] = w0; sm.W1[wave] = w1;
    }
    __syncthreads();
    if (tid == 0) {                                              // level 2: ordered walk
        double acc = 0.0;
        const int L = sm.L, n = sm.n;
        for (int w = 0; w < NWAVE; ++w) {
            sm.win[w] = acc;
            const int wbv = sm.wb[w];
            if (wbv == B_EMPTY) { sm.wfast[w] = 1; continue; }
            bool fast = false;
            if (wbv != B_INVALID && acc > 0.0 && expo_f64(acc) == wbv) {
                const double cand = acc + (parity_f64(acc) ? sm.W1[w] : sm.W0[w]);
                if (cand < pow2_f64(wbv + 1)) { acc = cand; fast = true; }
            }
            sm.wfast[w] = fast ? 1 : 0;
            if (fast) continue;
            for (int j = 0; j < 64; ++j) {
                const int t = w * 64 + j;
                sm.acc_in[t] = acc;
                const int bj = sm.bexp[t];
                if (bj == B_EMPTY) continue;
                bool done = false;
                if (bj != B_INVALID && acc > 0.0 && expo_f64(acc) == bj) {
                    const double cand = acc + (parity_f64(acc) ? sm.D1[t] : sm.D0[t]);
                    if (cand < pow2_f64(bj + 1)) { acc = cand; done = true; }
                }
                if (!done) {
                    const int wh = sm.has_copy[t];
                    if (!wh) { sm.flag = 1; continue; }
                    const int f0 = t * L, c = n - f0 < L ? n - f0 : L;
                    if (wh != 255) { const double* src = sm.pool[wh - 1]; for (int k = 0; k < c; ++k) acc = acc + src[k]; }   // the chunk's true chain
                    else for (int k = 0; k < c; ++k) acc = acc + scratch[f0 + k];           // pool exhausted (rare): global copy
                }
            }
        }
        sm.total = acc;
    }
    __syncthreads();
    const double total = sm.total;
    if (want_acc_in) {
        if (lane == 0 && sm.wfast[wave] && sm.wb[wave] != B_EMPTY) {   // level 1': replay skipped waves from their exact start
            double acc = sm.win[wave];
            for (int j = 0; j < 64; ++j) {
                const int t = wave * 64 + j;
                sm.acc_in[t] = acc;
                if (sm.bexp[t] == B_EMPTY) continue;
                acc = acc + (parity_f64(acc) ? sm.D1[t] : sm.D0[t]);
            }
        }
        __syncthreads();
        *acc_in_out = sm.acc_in[tid];
    }
    __syncthreads();
    return total;
}

// DECODE = 0: coder interval of sym[row] for `rows` rows (grid-stride over rows).
// DECODE = 1: one block per stream: decode a symbol, update the device-resident decoder state.
template <int DECODE>
__global__ __launch_bounds__(T_PAR) void cdf_par_kernel(const float* __restrict__ logits, int ld, int rows, int V,
                                                        const uint32_t* __restrict__ sym, uint32_t* __restrict__ c_lo,
                                                        uint32_t* __restrict__ c_hi, double* __restrict__ scratch_all,
                                                        cz_dec_state* __restrict__ dst, uint32_t* __restrict__ sym_out,
                                                        unsigned long long step_i, unsigned* __restrict__ fallback_count,
                                                        const unsigned long long* __restrict__ step_dev) {
    __shared__ par_smem sm;
    if (DECODE && step_dev) step_i = *step_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* scratch = scratch_all + (size_t)blockIdx.x * V;
    const int L = (V + T_PAR - 1) / T_PAR;
    const int first = tid * L;
    const int cnt = first >= V ? 0 : (V - first < L ? V - first : L);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        cz_dec_state d;
        if (DECODE) {
            d = dst[row];
            if (!d.active || step_i >= d.token_count) {       // uniform across the block
                if (tid == 0 && d.active) { d.active = 0; dst[row] = d; }
                continue;
            }
        }
        const float* l = logits + (size_t)row * ld;
        if (tid == 0) { sm.flag = 0; sm.L = L; sm.n = V; sm.found = (unsigned)V - 1u; sm.ok = 1; }
        float ef[LMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cnt) { ef[k] = l[first + k]; if (ef[k] > m) m = ef[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { float y = __shfl_xor(m, o); m = y > m ? y : m; }
        if (lane == 0) sm.redf[wave] = m;
        __syncthreads();
        m = sm.redf[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m = sm.redf[w] > m ? sm.redf[w] : m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cnt) ef[k] = cz_expf(ef[k] - m);
        double csum_e = 0.0;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cnt) csum_e += (double)ef[k];
        const double sum = exact_seq_sum([&](int k) { return (double)ef[k]; }, csum_e, cnt, first, sm, scratch, false, nullptr);
        bool fallback = sm.flag != 0 || !(sum > 0.0);            // sum <= 0 (uniform pdf, :794) or NaN: take the serial walk
        double acc = 0.0;
        // pdf term k of this thread, exactly as the reference computes it (:799): exps[i] / sum  (recomputed, not stored: registers)
        auto pd = [&](int k) { return (double)ef[k] / sum; };
        if (!fallback) {
            exact_seq_sum(pd, csum_e / sum, cnt, first, sm, scratch, true, &acc);
            fallback = sm.flag != 0;
        }
        if (fallback) {                                          // rare: exact serial walk by one lane
            if (tid == 0) {
                if (fallback_count) atomicAdd(fallback_count, 1u);
                if (DECODE) {
                    uint32_t s = 0;
                    if (serial_decode_row(l, V, 0, d, &s)) { d.out_ids[step_i + 1] = s; sym_out[row] = s; }
                    else { d.active = 0; sym_out[row] = 0; }
                    dst[row] = d;
                } else {
                    uint32_t lo, hi;
                    serial_pairs_row(l, V, 0, sym[row], &lo, &hi);
                    c_lo[row] = lo; c_hi[row] = hi;
                }
            }
            __syncthreads();
            continue;
        }
        const double acc_in = acc;
        if (!DECODE) {
            const uint32_t s = sym[row];
            if (tid == 0 && s == 0) c_lo[row] = 0;
            if (tid == 0 && s >= (uint32_t)V) { c_lo[row] = 0; c_hi[row] = 0; }
#pragma unroll
            for (int k = 0; k < LMAX; ++k)
                if (k < cnt) {
                    acc = acc + pd(k);
                    const uint32_t i = (uint32_t)(first + k);
                    if (i == s || i + 1 == s) {
                        const uint32_t v = (i + 1 == (uint32_t)V) ? CDF_TOTAL_U : quantize_u(acc);   // cdf[n] = total forced (:822)
                        if (i == s) c_hi[row] = v; else c_lo[row] = v;
                    }
                }
        } else {
            if (tid == 0) { bool ok; sm.value = dec_value(d, ok); sm.ok = ok ? 1 : 0; }
            __syncthreads();
            if (!sm.ok) {                                        // the reference would panic (range == 0)
                if (tid == 0) { d.active = 0; dst[row] = d; sym_out[row] = 0; }
                __syncthreads();
                continue;
            }
            const uint32_t value = sm.value;
            // smallest index whose upper cdf bound exceeds value (== the reference's binary search, :508-514)
            unsigned mine = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < LMAX; ++k)
                if (k < cnt) {
                    acc = acc + pd(k);
                    const uint32_t i = (uint32_t)(first + k);
                    const uint32_t v = (i + 1 == (uint32_t)V) ? CDF_TOTAL_U : quantize_u(acc);
                    if (v > value && mine == 0xffffffffu) mine = i;
                }
            if (mine != 0xffffffffu) atomicMin(&sm.found, mine);
            __syncthreads();
            const uint32_t s = sm.found;
            if (s >= (uint32_t)first && s < (uint32_t)(first + cnt)) {      // owner of s provides cdf[s], cdf[s+1]
                double a = acc_in; uint32_t lo = quantize_u(acc_in), hi = 0;
                if (s == 0) lo = 0;
#pragma unroll
                for (int k = 0; k < LMAX; ++k)
                    if (k < cnt) {
                        a = a + pd(k);
                        const uint32_t i = (uint32_t)(first + k);
                        const uint32_t v = (i + 1 == (uint32_t)V) ? CDF_TOTAL_U : quantize_u(a);
                        if (i + 1 == s) lo = v;
                        if (i == s) hi = v;
                    }
                sm.lo = lo; sm.hi = hi;
            }
            __syncthreads();
            if (tid == 0) {
                dec_update(d, sm.lo, sm.hi);
                d.out_ids[step_i + 1] = s;
                sym_out[row] = s;
                dst[row] = d;
            }
        }
        __syncthreads();
    }
}

static double* g_scratch = nullptr; static size_t g_scratch_bytes = 0;
static hipError_t ensure_scratch(size_t bytes) {
    if (bytes <= g_scratch_bytes) return hipSuccess;
    if (g_scratch) { hipError_t e = hipFree(g_scratch); (void)e; g_scratch = nullptr; g_scratch_bytes = 0; }
    hipError_t e = hipMalloc((void**)&g_scratch, bytes);
    if (e == hipSuccess) g_scratch_bytes = bytes;
    return e;
}
bool cz_cdf_par_supported(int vocab, int mode) { return mode == 0 && vocab <= T_PAR * LMAX; }

hipError_t cz_launch_cdf_pairs(hipStream_t st, const float* logits, int ld, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi,
                               int rows, int vocab, int mode, unsigned* fallback_count) {
    if (rows <= 0) return hipSuccess;
    if (!cz_cdf_par_supported(vocab, mode)) return cz_launch_cdf_pairs_serial(st, logits, ld, sym, c_lo, c_hi, rows, vocab, mode);
    const int grid = rows < 1024 ? rows : 1024;
    hipError_t e = ensure_scratch((size_t)1024 * vocab * sizeof(double));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cdf_par_kernel<0>, dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, c_lo, c_hi, g_scratch, nullptr, nullptr, 0ull,
                       fallback_count, nullptr);
    return hipGetLastError();
}
hipError_t cz_launch_decode_symbol(hipStream_t st, const float* logits, int ld, cz_dec_state* st_arr, uint32_t* sym_out, int nstreams,
                                   int vocab, int mode, unsigned long long i, unsigned* fallback_count, const unsigned long long* step_dev) {
    if (nstreams <= 0) return hipSuccess;
    if (!cz_cdf_par_supported(vocab, mode)) return cz_launch_decode_symbol_serial(st, logits, ld, st_arr, sym_out, nstreams, vocab, mode, i, step_dev);
    hipError_t e = ensure_scratch((size_t)1024 * vocab * sizeof(double));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cdf_par_kernel<1>, dim3(nstreams), dim3(T_PAR), 0, st, logits, ld, nstreams, vocab, nullptr, nullptr, nullptr, g_scratch, st_arr,
                       sym_out, i, fallback_count, step_dev);
    return hipGetLastError();
}
// allocate the scratch up front (graph capture forbids hipMalloc inside the captured region)
hipError_t cz_cdf_par_prepare(int vocab) { return ensure_scratch((size_t)1024 * vocab * sizeof(double)); }

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
#include <cstddef>
#include "cz_cdf_common.h"

#define T_PAR 1024
#define LMAX_SMOLLM 48                 // 1024 x 48 = 49 152 symbols (SmolLM2)
#define LMAX_RWKV 68                   // 1024 x 68 >= 65 536 + 256 symbols (RWKV World vocabulary + literal bytes), multiple of 4
#define NWAVE (T_PAR / 64)
#define POOL_SLOTS 48
#define CDF_STAGE_FLOATS 12288          // 48 KB at the start of par_smem_t double as the staging area of a row's logits (cdf_par_kernel's load phase)
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

template <int LMAX> struct par_smem_t {
    double sD0[T_PAR], sD1[T_PAR];   // inclusive segmented scan of the chunk transducers (delta for start parity 0 / 1)
    double segacc[T_PAR];            // exact acc at the start of the segment headed by chunk t; later: exact acc_in of chunk t
    int bexp[T_PAR];                 // binade of a chunk's transducer, B_INVALID (no transducer) or B_EMPTY
    unsigned char cls[T_PAR];        // 1 = provably clean (no copy kept), 0 = other
    unsigned char sp[T_PAR];         // parities of sD0 / sD1 (bit 0 / bit 1) on the binade's grid
    unsigned char has_copy[T_PAR];   // 0 = none, 1..POOL = LDS pool slot + 1, 255 = global scratch
    short heads[T_PAR];              // compacted list of segment-head chunks
    short headof[T_PAR];             // head chunk of the segment a chunk belongs to
    double pool[POOL_SLOTS][LMAX];   // copies of the terms of chunks that are not provably clean (usually < 30 per row)
    unsigned pool_next;
    double wsum[NWAVE];
    double wD0[NWAVE], wD1[NWAVE]; int wp[NWAVE], wflag[NWAVE], wcnt[NWAVE], whead[NWAVE];
    double total;
    float redf[NWAVE];
    int flag, L, n, nheads;
    unsigned found;
    uint32_t value, lo, hi;
    int ok;
};

// Lane shuffles with the lane index passed in.  The kernels launder their thread index through an empty asm once per row, so the
// permute addresses (and everything else derived from the thread index) are NOT loop invariants: hoisted out of the row loop they
// stay live across the whole body — ~45 VGPRs and ~100 SGPRs that the 128-register budget of a 1024-thread block then spills.
__device__ __forceinline__ int shfl_up_i(int v, int o, int lane) { return __builtin_amdgcn_ds_bpermute(((lane - o) & 63) << 2, v); }   // callers guard lane >= o
__device__ __forceinline__ double shfl_up_d(double v, int o, int lane) {
    const int a = ((lane - o) & 63) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}
__device__ __forceinline__ float shfl_xor_f(float v, int o, int lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ o) << 2, __float_as_int(v))); }
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }

template <class SM> __device__ __forceinline__ double block_excl_scan(double v, SM& sm, int tid) {     // approximate (any order): only guesses binades
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { double y = shfl_up_d(inc, o, lane); if (lane >= o) inc += y; }
    if (lane == 63) sm.wsum[wave] = inc;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += sm.wsum[w];
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {       // l uniform (v_readlane_b32 x 2)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// transducer composition "f then g" on one binade grid: out.D_p = f.D_p + g.D_{p xor parity(f.D_p)}
struct tdx { double d0, d1; int p; };      // p: bit0 = parity(d0/u), bit1 = parity(d1/u)
__device__ __forceinline__ tdx td_compose(const tdx& f, const tdx& g) {
    const int q0 = f.p & 1, q1 = 1 ^ ((f.p >> 1) & 1);
    tdx h;
    h.d0 = f.d0 + (q0 ? g.d1 : g.d0);
    h.d1 = f.d1 + (q1 ? g.d1 : g.d0);
    const int g0 = g.p & 1, g1 = (g.p >> 1) & 1;
    h.p = ((f.p & 1) ^ (q0 ? g1 : g0)) | ((((f.p >> 1) & 1) ^ (q1 ? g1 : g0)) << 1);
    return h;
}

// tuning aid: block 0 / thread 0 accumulates wall_clock64 ticks (10 ns) per phase when CZ_CDF_PHASE_TIMERS is defined
#ifdef CZ_CDF_PHASE_TIMERS
__device__ unsigned long long g_cdf_phase[16];
#define PH(i) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_cdf_phase[i], t_ - ph_t); ph_t = t_; } } while (0)
#else
#define PH(i) do { } while (0)
#endif
// Exact left-to-right sum of the block's terms xf(k) (thread t owns terms first..first+cnt-1, k < cnt <= LMAX).
// Returns the exact total; with want_acc_in also the exact running sum before this thread's first term.
// csum = APPROXIMATE sum of this thread's terms (any rounding; it only feeds the binade guesses).
template <int LMAX, typename XF>
__device__ __forceinline__ double exact_seq_sum(XF xf, double csum, int cnt, int first, par_smem_t<LMAX>& sm, double* scratch, bool want_acc_in,
                                                double* acc_in_out, unsigned long long& ph_t, int tid) {
    static_assert(LMAX <= 128, "the resolve holds a chunk's terms in two registers per lane");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) sm.pool_next = 0;
    const double pstart = block_excl_scan(csum, sm, tid);            // (contains barriers: the reset above is visible afterwards)
    const double pend = pstart + csum;
    PH(8);
    int b = B_INVALID; double d0 = 0.0, d1 = 0.0;
    // Whether this chunk must keep a copy of its terms is known BEFORE its chain runs: inside the margin below the approximate
    // prefix (within ~n*2^-53 < 1e-11 relative of the true one) proves that the true acc stays in binade bs across the whole chunk
    // (and then r0 + sum <= pend < top, so the transducer exists).  Every other non-empty chunk parks its terms while the chain
    // is evaluated — one evaluation of xf per term, which matters when a term costs three f64 divisions.
    const bool in_range = cnt > 0 && pstart > 1e-290 && pend < 1e290;
    const int bs = in_range ? expo_f64(pstart) : 0;
    const double r0 = pow2_f64(bs), u = pow2_f64(bs - 52), r1 = r0 + u, top = pow2_f64(bs + 1);
    const bool copy = cnt > 0 && !(in_range && pstart > r0 * (1.0 + 1e-9) && pend < top * (1.0 - 1e-9));
    unsigned char where = 0;
    double* park = nullptr;
    if (copy) {
        const unsigned slot = atomicAdd(&sm.pool_next, 1u);
        if (slot < POOL_SLOTS) { where = (unsigned char)(slot + 1); park = sm.pool[slot]; }
        else { where = 255; park = scratch + first; }                                  // pool exhausted (rare): global copy
    }
    if (cnt == 0) b = B_EMPTY;
    else {
        double e0 = r0, e1 = r1;
#pragma unroll
        for (int k = 0; k < LMAX; ++k)
            if (k < cnt) {
                const double x = xf(k);
                e0 = e0 + x; e1 = e1 + x;
                if (copy) park[k] = x;
            }
        if (in_range && e0 < top && e1 < top) { b = bs; d0 = e0 - r0; d1 = e1 - r1; }
    }
    PH(9);
    const bool clean = (b != B_INVALID && b != B_EMPTY && !copy);
    sm.bexp[tid] = b; sm.has_copy[tid] = where; sm.cls[tid] = clean ? 1 : 0;
    __threadfence_block();
    __syncthreads();
    PH(10);
    // ---- segments: maximal runs of provably-clean chunks of one binade; every other non-empty chunk is a singleton
    const bool nonempty = b != B_EMPTY;
    bool head = nonempty;
    if (clean && tid > 0 && sm.cls[tid - 1] && sm.bexp[tid - 1] == b) head = false;
    tdx v; v.d0 = d0; v.d1 = d1; v.p = 0;
    if (b != B_INVALID && b != B_EMPTY) {
        const double inv_u = pow2_f64(52 - b);
        v.p = (int)((long long)(d0 * inv_u) & 1) | ((int)((long long)(d1 * inv_u) & 1) << 1);
    }
    // wave-level segmented inclusive scan (flag = "a head lies inside the range covered so far")
    int flag = head ? 1 : 0;
    int hidx = head ? tid : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        tdx u; u.d0 = shfl_up_d(v.d0, o, lane); u.d1 = shfl_up_d(v.d1, o, lane); u.p = shfl_up_i(v.p, o, lane);
        const int uf = shfl_up_i(flag, o, lane), uh = shfl_up_i(hidx, o, lane);
        if (lane >= o) {
            if (!flag) v = td_compose(u, v);
            flag |= uf;
            hidx = hidx > uh ? hidx : uh;
        }
    }
    // head count prefix (for compaction)
    int hc = head ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = shfl_up_i(hc, o, lane); if (lane >= o) hc += y; }
    if (lane == 63) { sm.wD0[wave] = v.d0; sm.wD1[wave] = v.d1; sm.wp[wave] = v.p; sm.wflag[wave] = flag; sm.wcnt[wave] = hc; sm.whead[wave] = hidx; }
    __syncthreads();
    {
        // carry from the preceding waves (serial over <= NWAVE entries, every thread for itself: cheap, no extra barrier)
        tdx carry; carry.d0 = 0.0; carry.d1 = 0.0; carry.p = 0; int have = 0, hbase = 0, chead = -1;
        for (int w = 0; w < wave; ++w) {
            tdx t; t.d0 = sm.wD0[w]; t.d1 = sm.wD1[w]; t.p = sm.wp[w];
            if (sm.wflag[w]) { carry = t; have = 1; } else if (have) carry = td_compose(carry, t);
            hbase += sm.wcnt[w];
            if (sm.whead[w] > chead) chead = sm.whead[w];
        }
        if (!flag && have) v = td_compose(carry, v);
        if (hidx < 0) hidx = chead;
        if (head) sm.heads[hbase + hc - 1] = (short)tid;
        if (tid == T_PAR - 1) sm.nheads = hbase + hc;
    }
    sm.sD0[tid] = v.d0; sm.sD1[tid] = v.d1; sm.sp[tid] = (unsigned char)v.p; sm.headof[tid] = (short)hidx;
    __syncthreads();
    PH(11);
    // ---- ordered resolve over the segments (wave 0, every lane runs the same scalar walk; segment descriptors and the terms of
    //      a dirty chunk are fetched by the lanes in parallel and fed to the chain with v_readlane, not with dependent LDS reads)
    if (wave == 0) {
        double acc = 0.0;
        const int L = sm.L, n = sm.n, nh = sm.nheads, nchunks = (n + L - 1) / L;
        for (int base = 0; base < nh; base += 64) {
            // lane j fetches everything the walk needs about segment base + j; the walk itself then reads registers only
            const int i = base + lane;
            int t = 0, bj = B_INVALID, wh = 0; double D0 = 0.0, D1 = 0.0;
            if (i < nh) {
                t = sm.heads[i];
                const int te = (i + 1 < nh ? (int)sm.heads[i + 1] : nchunks) - 1;         // last chunk of the segment
                bj = sm.bexp[t]; wh = sm.has_copy[t]; D0 = sm.sD0[te]; D1 = sm.sD1[te];
            }
            double myacc = 0.0;
            const int lim = nh - base < 64 ? nh - base : 64;
            for (int j = 0; j < lim; ++j) {
                if (lane == j) myacc = acc;
                const int bjj = __builtin_amdgcn_readlane(bj, j);
                bool done = false;
                if (bjj != B_INVALID && acc > 0.0 && expo_f64(acc) == bjj) {
                    const double cand = acc + (parity_f64(acc) ? readlane_f64(D1, j) : readlane_f64(D0, j));
                    if (cand < pow2_f64(bjj + 1)) { acc = cand; done = true; }
                }
                if (!done) {
                    const int whj = __builtin_amdgcn_readlane(wh, j), tj = __builtin_amdgcn_readlane(t, j);
                    if (!whj) { if (lane == 0) sm.flag = 1; continue; }   // a "provably clean" segment failed its check: serial fallback
                    const int f0 = tj * L, c = n - f0 < L ? n - f0 : L;
                    const double* src = whj != 255 ? sm.pool[whj - 1] : scratch + f0;     // LDS pool slot, or (pool exhausted, rare) the global copy
                    const double x0 = lane < c ? src[lane] : 0.0;
                    const double x1 = (LMAX > 64 && 64 + lane < c) ? src[64 + lane] : 0.0;
                    // the chunk's true chain; lanes past the chunk hold +0.0 (acc + 0.0 == acc), so the trip count is a compile-time
                    // constant and the v_readlane pairs are scheduled ahead of the dependent adds
#pragma unroll
                    for (int k = 0; k < (LMAX < 64 ? LMAX : 64); ++k) acc = acc + readlane_f64(x0, k);
#pragma unroll
                    for (int k = 64; k < LMAX; ++k) acc = acc + readlane_f64(x1, k - 64);
                }
            }
            if (i < nh) sm.segacc[t] = myacc;
        }
        if (lane == 0) sm.total = acc;
    }
    __syncthreads();
    PH(12);
    const double total = sm.total;
    if (want_acc_in) {
        double a = 0.0;
        if (nonempty) {
            const int h = sm.headof[tid];
            a = sm.segacc[h];
            if (h != tid) a = a + (parity_f64(a) ? sm.sD1[tid - 1] : sm.sD0[tid - 1]);   // exclusive prefix inside the clean segment
        }
        *acc_in_out = a;
    }
    __syncthreads();
    return total;
}

// DECODE = 0: coder interval of sym[row] for `rows` rows (grid-stride over rows).
// DECODE = 1: one block per stream: decode a symbol, update the device-resident decoder state.
// MODE = 0: SmolLM (softmax_pdf + quantize_pdf_to_cdf, two chains over V terms).
// MODE = 1: RWKV with literals (softmax_pdf_floor + combined_pdf_with_literals + the inline quantiser, src/main.rs:758-782,
//           2303-2325): FOUR sequential f64 chains — sum of exps (V), sum of floored probabilities (V), sum of the scaled
//           distribution with its 256 literal entries (V+256), running cdf (V+256) — each evaluated exactly in parallel.
// FULL: the alphabet fills the block exactly (V == T_PAR * LMAX, e.g. 49 152 = 1024 x 48): every `k < cnt` guard folds away, and with
// it the exec masks the unrolled loops would otherwise keep alive (hundreds of spilled SGPRs, ~40 MB of scratch traffic per launch)
template <int DECODE, int MODE, int LMAX, bool FULL>
__global__ __launch_bounds__(T_PAR) void cdf_par_kernel(const float* __restrict__ logits, int ld, int rows, int V,
                                                        const uint32_t* __restrict__ sym, uint32_t* __restrict__ c_lo,
                                                        uint32_t* __restrict__ c_hi, double* __restrict__ scratch_all,
                                                        cz_dec_state* __restrict__ dst, uint32_t* __restrict__ sym_out,
                                                        unsigned long long step_i, unsigned* __restrict__ fallback_count,
                                                        const unsigned long long* __restrict__ step_dev) {
    __shared__ par_smem_t<LMAX> sm;
    static_assert(offsetof(par_smem_t<LMAX>, pool_next) >= CDF_STAGE_FLOATS * sizeof(float), "the staging area must end before the fields that live across the load phase (flag, L, n, redf ...)");
    if (DECODE && step_dev) step_i = *step_dev;
    if (!DECODE && step_dev) { c_lo += *step_dev * (unsigned long long)rows; c_hi += *step_dev * (unsigned long long)rows; }   // lock-step encode under graph replay
    const int n = MODE ? V + 256 : V;                              // symbols of the coded alphabet
    double* scratch = scratch_all + (size_t)blockIdx.x * n;
    int L = (n + T_PAR - 1) / T_PAR;
    if (L >= 8) L = (L + 3) & ~3;                                  // chunks start on 16-B boundaries: 4 logits per load
    if (FULL) L = LMAX;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // see shfl_up_i
        const int first = tid * L;
        const int cntA = FULL ? LMAX : (first >= V ? 0 : (V - first < L ? V - first : L));       // owned vocabulary entries (terms of the chains over V)
        const int cnt = FULL ? LMAX : (first >= n ? 0 : (n - first < L ? n - first : L));        // owned symbols (vocabulary, then literals)
        if (DECODE) {                                         // only thread 0 ever holds the whole decoder state (it lives in memory, not in 1024 x 24 registers)
            const unsigned active = dst[row].active;
            if (!active || step_i >= dst[row].token_count) {  // uniform across the block
                if (tid == 0 && active) dst[row].active = 0;
                continue;
            }
        }
        const float* l = logits + (size_t)row * ld;
        if (tid == 0) { sm.flag = 0; sm.L = L; sm.n = V; sm.found = (unsigned)n - 1u; sm.ok = 1; }
        unsigned long long ph_t = 0;
#ifdef CZ_CDF_PHASE_TIMERS
        ph_t = wall_clock64();
#endif
        float ef[LMAX];
        float m = -INFINITY;
        // Every load unconditional, from a clamped address (entries past the vocabulary are read but never used): a load inside a
        // per-lane `k < cnt` branch is waited for on the spot, i.e. up to LMAX dependent round trips per row.
        if (MODE == 1 && (L & 3) == 0 && ((size_t)l & 15) == 0 && (V & 3) == 0 && V >= 4) {
            // (MODE 1 = the RWKV alphabets, 68 logits per thread: 8 192 rows 7.30 -> 5.38 ms, one decode row 198 -> 169 us.  The SmolLM alphabet,
            // 48 per thread, LOSES with it — 2.32 -> 2.39 ms on the general instantiation, 2.89 on the FULL one — and keeps the direct loads below;
            // profiles/r03_cdf_staged_load_kernel_times.txt.)
            // A thread owns L CONSECUTIVE logits, so loading them straight from memory has the 64 lanes of a wave touch 64 different lines per
            // instruction, and the block's 1024 x (L * 4 B) working set thrashes the 32 KB L1: a row's 192-256 KB took 27-56 us to arrive
            // (profiles/r03_cdf_par_phases.txt).  Instead the block streams the row COALESCED (one KB per wave instruction) through the LDS that
            // the chains use later — dead between rows —, a few waves' chunks per round, and every thread picks its chunk up from there.
            // Chunks lie LP4 float4 apart with LP4 odd, so the sixteen lanes of a ds_read_b128 phase hit distinct banks.
            const int L4 = L >> 2, LP4 = (L4 & 1) ? L4 : L4 + 1;
            int RW = (CDF_STAGE_FLOATS / 4) / (64 * LP4);                     // waves whose chunks one round stages
            RW = RW < 1 ? 1 : (RW > NWAVE ? NWAVE : RW);
            const int per4 = RW * 64 * L4, V4 = V >> 2, rounds = (NWAVE + RW - 1) / RW;
            float4* stage4 = reinterpret_cast<float4*>(&sm);
            const float4* l4 = reinterpret_cast<const float4*>(l);
            float4 pf[3];                                                      // (per4 <= CDF_STAGE_FLOATS / 4 = 3 x 1024)
            auto issue = [&](const int r) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { int g = r * per4 + tid + T_PAR * i; g = g < V4 ? g : V4 - 1; pf[i] = l4[g]; }   // clamped, never conditional
            };
            issue(0);
            // The staging area overlays fields of `sm` that the PREVIOUS row of this block reads until its very end (DECODE: sm.bexp[tid + 1] in the
            // symbol search): no wave may write it before every wave has left that row.  Every path through the row loop ends in a barrier
            // today; this one makes the overlay safe by construction (one barrier per row, under the first round's loads).
            __syncthreads();
#pragma unroll 1
            for (int r = 0; r < rounds; ++r) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int f4 = tid + T_PAR * i;
                    if (f4 < per4) {
                        int dst = f4;
                        if (LP4 != L4) { const int q = f4 / L4; dst = q * LP4 + (f4 - q * L4); }
                        stage4[dst] = pf[i];
                    }
                }
                if (r + 1 < rounds) issue(r + 1);                              // in flight across the two barriers below
                __syncthreads();
                if (wave / RW == r) {
                    const float4* mine = stage4 + (tid - r * RW * 64) * LP4;
#pragma unroll
                    for (int k = 0; k < LMAX; k += 4)
                        if (k < L) { const float4 q = mine[k >> 2]; ef[k] = q.x; ef[k + 1] = q.y; ef[k + 2] = q.z; ef[k + 3] = q.w; }
                }
                __syncthreads();
            }
        } else if ((L & 3) == 0 && ((size_t)l & 15) == 0 && (V & 3) == 0 && V >= 4) {   // a lane's chunk is contiguous: 16-B loads (a quarter of the cache-line visits)
#pragma unroll
            for (int k = 0; k < LMAX; k += 4) {
                const int at = FULL ? first + k : (first + k < V - 4 ? first + k : V - 4);
                const float4 q = *reinterpret_cast<const float4*>(l + at);
                ef[k] = q.x; ef[k + 1] = q.y; ef[k + 2] = q.z; ef[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LMAX; ++k) ef[k] = l[first + k < V ? first + k : V - 1];
        }
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cntA && ef[k] > m) m = ef[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { float y = shfl_xor_f(m, o, lane); m = y > m ? y : m; }
        if (lane == 0) sm.redf[wave] = m;
        __syncthreads();
        m = sm.redf[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m = sm.redf[w] > m ? sm.redf[w] : m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cntA) ef[k] = cz_expf(ef[k] - m);
        double csum_e = 0.0;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cntA) csum_e += (double)ef[k];
        PH(0);
        const double sum = exact_seq_sum<LMAX>([&](int k) { return (double)ef[k]; }, csum_e, cntA, first, sm, scratch, false, nullptr, ph_t, tid);
        PH(1);
        bool fallback = sm.flag != 0 || !(sum > 0.0);            // sum <= 0 (uniform pdf, :794) or NaN: take the serial walk
        double acc = 0.0;
        // MODE 0: pdf term k of this thread, exactly as the reference computes it (:799): exps[i] / sum (recomputed, not stored: registers)
        // MODE 1: the same through the floor (:763), the renormalisation (:765), the literal mix (:770-780)
        double norm = 1.0, sum2 = 0.0;
        const double p_escape_total = 256.0 * P_MIN;             // :771
        const double scale = (1.0 - p_escape_total) > 0.0 ? (1.0 - p_escape_total) : 0.0;
        const double lit = p_escape_total > 0.0 ? p_escape_total / 256.0 : 0.0;
        // every division of a row is by one of three row-wide divisors: div_by() with their correctly rounded reciprocals (cz_cdf_common.h: same bits)
        const double inv_sum = 1.0 / sum;
        double inv_norm = 1.0, inv_sum2 = 1.0;
        auto fl = [&](int k) { const double p = div_by((double)ef[k], sum, inv_sum); return p > P_MIN ? p : P_MIN; };   // (exps/sum).max(p_floor)
        auto mix = [&](int k) { return k < cntA ? div_by(fl(k), norm, inv_norm) * scale : lit; };                       // base[i] /= norm; *= scale | literal
        auto pd = [&](int k) {
            if (MODE == 0) return div_by((double)ef[k], sum, inv_sum);
            const double x = mix(k);
            return sum2 > 0.0 ? div_by(x, sum2, inv_sum2) : x;                                                       // :779
        };
        if (MODE == 1 && !fallback) {
            // approximate chunk sums (they only steer the binade guesses): reciprocal multiply, no correction steps
            double cs = 0.0;
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cntA) { const double p = (double)ef[k] * inv_sum; cs += p > P_MIN ? p : P_MIN; }
            norm = exact_seq_sum<LMAX>(fl, cs, cntA, first, sm, scratch, false, nullptr, ph_t, tid);
            PH(2);
            fallback = sm.flag != 0 || !(norm > 0.0);
            if (!fallback) {
                if (tid == 0) sm.n = n;                          // the last two chains run over vocabulary + literals
                inv_norm = 1.0 / norm;
                const double cs2 = cs * (scale / norm) + (double)(cnt - cntA) * lit;
                sum2 = exact_seq_sum<LMAX>(mix, cs2, cnt, first, sm, scratch, false, nullptr, ph_t, tid);
                PH(3);
                fallback = sm.flag != 0 || !(sum2 > 0.0);
                if (!fallback) {
                    inv_sum2 = 1.0 / sum2;
                    exact_seq_sum<LMAX>(pd, cs2 / sum2, cnt, first, sm, scratch, true, &acc, ph_t, tid);
                    fallback = sm.flag != 0;
                }
            }
        }
        if (MODE == 0 && !fallback) {
            exact_seq_sum<LMAX>(pd, csum_e / sum, cnt, first, sm, scratch, true, &acc, ph_t, tid);
            fallback = sm.flag != 0;
        }
        if (fallback) {                                          // rare: exact serial walk by one lane
            if (tid == 0) {
                if (fallback_count) atomicAdd(fallback_count, 1u);
                if (DECODE) {
                    uint32_t s = 0;                          // (the state stays in memory: a local copy handed to a call would cost every lane a stack frame)
                    if (serial_decode_row(l, V, MODE, dst[row], &s)) { dst[row].out_ids[step_i + 1] = s; sym_out[row] = s; }
                    else { dst[row].active = 0; sym_out[row] = 0; }
                } else {
                    uint32_t lo, hi;
                    serial_pairs_row(l, V, MODE, sym[row], &lo, &hi);
                    c_lo[row] = lo; c_hi[row] = hi;
                }
            }
            __syncthreads();
            continue;
        }
        PH(4);
        const double acc_in = acc;
        // Only the chunk that holds the symbol walks its terms once more (exact acc_in in hand); the other fifteen waves are done.
        if (!DECODE) {
            const uint32_t s = sym[row];
            if (tid == 0 && s >= (uint32_t)n) { c_lo[row] = 0; c_hi[row] = 0; }
            if (cnt > 0 && s >= (uint32_t)first && s < (uint32_t)(first + cnt)) {
                const int srel = (int)s - first;                                        // the symbol's place inside this chunk
                uint32_t lo = s == 0 ? 0u : quantize_u(acc_in), hi = 0;                 // cdf[first] (value when s == first)
#pragma unroll
                for (int k = 0; k < LMAX; ++k)
                    if (k < cnt && k <= srel) {
                        acc = acc + pd(k);
                        const uint32_t v = quantize_u(acc);
                        if (k + 1 == srel) lo = v;
                        if (k == srel) hi = v;
                    }
                if (s + 1 == (uint32_t)n) hi = CDF_TOTAL_U;                              // cdf[n] = total forced (:822, :2321)
                c_lo[row] = lo; c_hi[row] = hi;
            }
        } else {
            if (tid == 0) { bool ok; sm.value = dec_value(dst[row], ok); sm.ok = ok ? 1 : 0; }
            sm.bexp[tid] = cnt > 0 ? (int)quantize_u(acc_in) : (int)CDF_TOTAL_U;        // cdf[first] of every chunk (bexp is free between chains)
            __syncthreads();
            if (!sm.ok) {                                        // the reference would panic (range == 0)
                if (tid == 0) { dst[row].active = 0; sym_out[row] = 0; }
                __syncthreads();
                continue;
            }
            const uint32_t value = sm.value;
            // the symbol = smallest index whose upper cdf bound exceeds value (== the reference's binary search, :508-514, which never
            // runs past n - 1).  cdf[first] is monotone over the chunks, so exactly one chunk brackets value.
            const uint32_t qstart = first == 0 ? 0u : (uint32_t)sm.bexp[tid];
            const uint32_t qend = tid + 1 < T_PAR ? (uint32_t)sm.bexp[tid + 1] : CDF_TOTAL_U;   // the last chunk ends at the forced total
            const bool last_chunk = cnt > 0 && first + cnt == n;
            if (cnt > 0 && qstart <= value && (value < qend || last_chunk)) {
                unsigned mine = 0xffffffffu;
                uint32_t prev = qstart, my_lo = 0, my_hi = 0;
#pragma unroll
                for (int k = 0; k < LMAX; ++k)
                    if (k < cnt) {
                        acc = acc + pd(k);
                        const bool forced = last_chunk && k + 1 == cnt;                 // cdf[n] = total forced (:822, :2321)
                        const uint32_t v = forced ? CDF_TOTAL_U : quantize_u(acc);
                        if ((v > value || forced) && mine == 0xffffffffu) { mine = (unsigned)k; my_lo = prev; my_hi = v; }   // cdf[i] <= value < cdf[i+1]
                        prev = v;
                    }
                sm.found = (unsigned)first + mine; sm.lo = my_lo; sm.hi = my_hi;
            }
            __syncthreads();
            const uint32_t s = sm.found;
            if (tid == 0) {
                cz_dec_state d = dst[row];
                dec_update(d, sm.lo, sm.hi);
                d.out_ids[step_i + 1] = s;
                sym_out[row] = s;
                dst[row] = d;
            }
        }
        __syncthreads();
        PH(5);
    }
}
#ifdef CZ_CDF_PHASE_TIMERS
extern "C" void cz_debug_cdf_phases(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cdf_phase), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_cdf_phase), z, sizeof z); }
}
#endif

// f4 — the probability the floor-softmax (softmax_pdf_floor, src/main.rs:758-766) gives sym[row]: p = max(e_s / S, floor) / norm with
// S = sum of exps and norm = sum of floored probabilities, both sequential f64 chains evaluated exactly in parallel as above.
// Consumer: cross_entropy_bits_over_span (src/main.rs:1725-1751); -log2 and the running sum stay on the host (libm).
template <int LMAX>
__global__ __launch_bounds__(T_PAR) void xe_prob_kernel(const float* __restrict__ logits, int ld, int rows, int V, const uint32_t* __restrict__ sym,
                                                        double* __restrict__ p_out, double* __restrict__ scratch_all, unsigned* __restrict__ fallback_count) {
    __shared__ par_smem_t<LMAX> sm;
    double* scratch = scratch_all + (size_t)blockIdx.x * V;
    int L = (V + T_PAR - 1) / T_PAR;
    if (L >= 8) L = (L + 3) & ~3;
    unsigned long long ph_t = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int first = tid * L;
        const int cnt = first >= V ? 0 : (V - first < L ? V - first : L);
        const float* l = logits + (size_t)row * ld;
        if (tid == 0) { sm.flag = 0; sm.L = L; sm.n = V; }
        float ef[LMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) ef[k] = l[first + k < V ? first + k : V - 1];      // (unconditional clamped loads, as in cdf_par_kernel)
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cnt && ef[k] > m) m = ef[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { float y = shfl_xor_f(m, o, lane); m = y > m ? y : m; }
        if (lane == 0) sm.redf[wave] = m;
        __syncthreads();
        m = sm.redf[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m = sm.redf[w] > m ? sm.redf[w] : m;
        __syncthreads();
        double csum_e = 0.0;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cnt) { ef[k] = cz_expf(ef[k] - m); csum_e += (double)ef[k]; }
        const double sum = exact_seq_sum<LMAX>([&](int k) { return (double)ef[k]; }, csum_e, cnt, first, sm, scratch, false, nullptr, ph_t, tid);
        bool fallback = sm.flag != 0 || !(sum > 0.0);
        const double inv_sum = 1.0 / sum;
        auto fl = [&](int k) { const double p = div_by((double)ef[k], sum, inv_sum); return p > P_MIN ? p : P_MIN; };
        double norm = 1.0;
        if (!fallback) {
            double cs = 0.0;
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cnt) { const double p = (double)ef[k] * inv_sum; cs += p > P_MIN ? p : P_MIN; }
            norm = exact_seq_sum<LMAX>(fl, cs, cnt, first, sm, scratch, false, nullptr, ph_t, tid);
            fallback = sm.flag != 0 || !(norm > 0.0);
        }
        const uint32_t s = sym[row];
        if (fallback) {
            if (tid == 0) {
                if (fallback_count) atomicAdd(fallback_count, 1u);
                pdf_ctx c;
                pdf_prepare(c, l, V, 1);
                p_out[row] = s < (uint32_t)V ? pdf_floor_at(c, (int)s) / c.norm : P_MIN;
            }
        } else {
            if (s >= (uint32_t)V) { if (tid == 0) p_out[row] = P_MIN; }                 // pdf.get(sym).unwrap_or(ac_p_min())
            else if (s >= (uint32_t)first && s < (uint32_t)(first + cnt)) {
                double p = 0.0;
#pragma unroll
                for (int k = 0; k < LMAX; ++k) if (first + k == (int)s) p = fl(k) / norm;
                p_out[row] = p;
            }
        }
        __syncthreads();
    }
}
// The same for the RWKV branch (cross_entropy_bits_over_span_rwkv, src/main.rs:1753-1787): p = combined_pdf_with_literals(logits)[sym] — the three
// sequential chains of cdf_par_kernel's MODE 1 (sum of exps over V, sum of floored probabilities over V, sum of the scaled distribution with its 256
// literal entries over V + 256), then the one entry; sym >= V + 256 -> ac_p_min() (`if idx < pdf.len() { pdf[idx] } else { ac_p_min() }`, :1779-1780).
template <int LMAX>
__global__ __launch_bounds__(T_PAR) void xe_prob_lit_kernel(const float* __restrict__ logits, int ld, int rows, int V, const uint32_t* __restrict__ sym,
                                                            double* __restrict__ p_out, double* __restrict__ scratch_all, unsigned* __restrict__ fallback_count) {
    __shared__ par_smem_t<LMAX> sm;
    const int n = V + 256;
    double* scratch = scratch_all + (size_t)blockIdx.x * n;
    int L = (n + T_PAR - 1) / T_PAR;
    if (L >= 8) L = (L + 3) & ~3;
    unsigned long long ph_t = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int tid = launder((int)threadIdx.x), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int first = tid * L;
        const int cntA = first >= V ? 0 : (V - first < L ? V - first : L);       // owned vocabulary entries
        const int cnt = first >= n ? 0 : (n - first < L ? n - first : L);        // owned symbols (vocabulary, then literals)
        const float* l = logits + (size_t)row * ld;
        if (tid == 0) { sm.flag = 0; sm.L = L; sm.n = V; }
        float ef[LMAX];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) ef[k] = l[first + k < V ? first + k : V - 1];      // (unconditional clamped loads, as in cdf_par_kernel)
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cntA && ef[k] > m) m = ef[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { float y = shfl_xor_f(m, o, lane); m = y > m ? y : m; }
        if (lane == 0) sm.redf[wave] = m;
        __syncthreads();
        m = sm.redf[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m = sm.redf[w] > m ? sm.redf[w] : m;
        __syncthreads();
        double csum_e = 0.0;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) if (k < cntA) { ef[k] = cz_expf(ef[k] - m); csum_e += (double)ef[k]; }
        const double sum = exact_seq_sum<LMAX>([&](int k) { return (double)ef[k]; }, csum_e, cntA, first, sm, scratch, false, nullptr, ph_t, tid);
        bool fallback = sm.flag != 0 || !(sum > 0.0);
        const double inv_sum = 1.0 / sum;
        double norm = 1.0, inv_norm = 1.0, sum2 = 0.0;
        const double p_escape_total = 256.0 * P_MIN;             // :771
        const double scale = (1.0 - p_escape_total) > 0.0 ? (1.0 - p_escape_total) : 0.0;
        const double lit = p_escape_total > 0.0 ? p_escape_total / 256.0 : 0.0;
        auto fl = [&](int k) { const double p = div_by((double)ef[k], sum, inv_sum); return p > P_MIN ? p : P_MIN; };
        auto mix = [&](int k) { return k < cntA ? div_by(fl(k), norm, inv_norm) * scale : lit; };
        if (!fallback) {
            double cs = 0.0;
#pragma unroll
            for (int k = 0; k < LMAX; ++k) if (k < cntA) { const double p = (double)ef[k] * inv_sum; cs += p > P_MIN ? p : P_MIN; }
            norm = exact_seq_sum<LMAX>(fl, cs, cntA, first, sm, scratch, false, nullptr, ph_t, tid);
            fallback = sm.flag != 0 || !(norm > 0.0);
            if (!fallback) {
                if (tid == 0) sm.n = n;                          // the last chain runs over vocabulary + literals
                inv_norm = 1.0 / norm;
                const double cs2 = cs * (scale / norm) + (double)(cnt - cntA) * lit;
                sum2 = exact_seq_sum<LMAX>(mix, cs2, cnt, first, sm, scratch, false, nullptr, ph_t, tid);
                fallback = sm.flag != 0;
            }
        }
        const uint32_t s = sym[row];
        if (fallback) {
            if (tid == 0) {
                if (fallback_count) atomicAdd(fallback_count, 1u);
                pdf_ctx c;
                pdf_prepare(c, l, V, 1);
                p_out[row] = s < (uint32_t)n ? pdf_at(c, (int)s) : P_MIN;
            }
        } else {
            if (s >= (uint32_t)n) { if (tid == 0) p_out[row] = P_MIN; }
            else if (s >= (uint32_t)first && s < (uint32_t)(first + cnt)) {
                double p = 0.0;
#pragma unroll
                for (int k = 0; k < LMAX; ++k)
                    if (first + k == (int)s) { const double x = mix(k); p = sum2 > 0.0 ? x / sum2 : x; }      // :779 (one division per row: the plain IEEE one)
                p_out[row] = p;
            }
        }
        __syncthreads();
    }
}
hipError_t cz_launch_xe_probs(hipStream_t st, const float* logits, int ld, const uint32_t* sym, double* p_out, int rows, int vocab, unsigned* fallback_count,
                              double* scratch, int mode) {
    if (rows <= 0) return hipSuccess;
    if (!scratch || vocab + (mode ? 256 : 0) > T_PAR * LMAX_RWKV) return hipErrorInvalidValue;
    const int grid = rows < CZ_CDF_PAR_MAX_GRID ? rows : CZ_CDF_PAR_MAX_GRID;
    if (mode) hipLaunchKernelGGL((xe_prob_lit_kernel<LMAX_RWKV>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, p_out, scratch, fallback_count);
    else if (vocab <= T_PAR * LMAX_SMOLLM) hipLaunchKernelGGL((xe_prob_kernel<LMAX_SMOLLM>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, p_out, scratch, fallback_count);
    else hipLaunchKernelGGL((xe_prob_kernel<LMAX_RWKV>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, p_out, scratch, fallback_count);
    return hipGetLastError();
}

bool cz_cdf_par_supported(int vocab, int mode) { return mode == 0 ? vocab <= T_PAR * LMAX_SMOLLM : vocab + 256 <= T_PAR * LMAX_RWKV; }

hipError_t cz_launch_cdf_pairs(hipStream_t st, const float* logits, int ld, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi,
                               int rows, int vocab, int mode, unsigned* fallback_count, double* g_scratch, const unsigned long long* out_step_dev) {
    if (rows <= 0) return hipSuccess;
    if (!cz_cdf_par_supported(vocab, mode)) return out_step_dev ? hipErrorInvalidValue : cz_launch_cdf_pairs_serial(st, logits, ld, sym, c_lo, c_hi, rows, vocab, mode);
    if (!g_scratch) return hipErrorInvalidValue;
    const int grid = rows < CZ_CDF_PAR_MAX_GRID ? rows : CZ_CDF_PAR_MAX_GRID;
    if (mode == 0 && vocab == T_PAR * LMAX_SMOLLM)
        hipLaunchKernelGGL((cdf_par_kernel<0, 0, LMAX_SMOLLM, true>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, c_lo, c_hi, g_scratch, nullptr, nullptr,
                           0ull, fallback_count, out_step_dev);
    else if (mode == 0)
        hipLaunchKernelGGL((cdf_par_kernel<0, 0, LMAX_SMOLLM, false>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, c_lo, c_hi, g_scratch, nullptr, nullptr,
                           0ull, fallback_count, out_step_dev);
    else
        hipLaunchKernelGGL((cdf_par_kernel<0, 1, LMAX_RWKV, false>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows, vocab, sym, c_lo, c_hi, g_scratch, nullptr, nullptr,
                           0ull, fallback_count, out_step_dev);
    return hipGetLastError();
}
int g_cdf_dec_grid_cap = 0;   // test knob (cz_debug_set "cdf_dec_grid_cap"): at most this many blocks per decode launch, so that a block owns SEVERAL streams
                              // (the row loop of the DECODE instantiations; 0 = one block per stream, what every real launch uses)
hipError_t cz_launch_decode_symbol(hipStream_t st, const float* logits, int ld, cz_dec_state* st_arr, uint32_t* sym_out, int nstreams,
                                   int vocab, int mode, unsigned long long i, unsigned* fallback_count, const unsigned long long* step_dev,
                                   double* g_scratch) {
    if (nstreams <= 0) return hipSuccess;
    if (!cz_cdf_par_supported(vocab, mode)) return cz_launch_decode_symbol_serial(st, logits, ld, st_arr, sym_out, nstreams, vocab, mode, i, step_dev);
    if (!g_scratch || nstreams > CZ_CDF_PAR_MAX_GRID) return hipErrorInvalidValue;
    const int rows_all = nstreams;                                            // (the kernels loop row = blockIdx.x, += gridDim.x over `rows`)
    const int grid = g_cdf_dec_grid_cap > 0 && g_cdf_dec_grid_cap < nstreams ? g_cdf_dec_grid_cap : nstreams;
    if (mode == 0 && vocab == T_PAR * LMAX_SMOLLM)
        hipLaunchKernelGGL((cdf_par_kernel<1, 0, LMAX_SMOLLM, true>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows_all, vocab, nullptr, nullptr, nullptr, g_scratch,
                           st_arr, sym_out, i, fallback_count, step_dev);
    else if (mode == 0)
        hipLaunchKernelGGL((cdf_par_kernel<1, 0, LMAX_SMOLLM, false>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows_all, vocab, nullptr, nullptr, nullptr, g_scratch,
                           st_arr, sym_out, i, fallback_count, step_dev);
    else
        hipLaunchKernelGGL((cdf_par_kernel<1, 1, LMAX_RWKV, false>), dim3(grid), dim3(T_PAR), 0, st, logits, ld, rows_all, vocab, nullptr, nullptr, nullptr, g_scratch,
                           st_arr, sym_out, i, fallback_count, step_dev);
    return hipGetLastError();
}
// per-session scratch (one spill row per resident block; only touched when a row exhausts the LDS pool)
size_t cz_cdf_par_scratch_bytes(int vocab, int mode) { return (size_t)CZ_CDF_PAR_MAX_GRID * (vocab + (mode ? 256 : 0)) * sizeof(double); }

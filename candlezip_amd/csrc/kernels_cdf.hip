// kernels_cdf.hip — logits -> pdf -> 2^30 integer CDF -> coder interval / decoded symbol, on the device.
//
// Restates, bit for bit, reference src/main.rs:
//   softmax_pdf 784-801, softmax_pdf_floor 758-767, combined_pdf_with_literals 769-782, quantize_pdf_to_cdf 805-824
//   (and the identical inline quantiser 2305-2321 / 2815-2827), ArithmeticDecoder::decode_symbol_counts 500-548,
//   ArithmeticDecoder::new / get_bit 419-449.
// The reference's f64 sums are strict left-to-right chains over the vocabulary (Rust never reassociates), and the
// CDF is floor(acc * 2^30) of such a chain, so a reordered parallel sum would change payload bits.
//
// *_serial kernels (this file, v1): one LANE walks one row's chain exactly as the reference does.  Throughput comes
// from rows >> lanes on encode; they are also the on-device reference for the exact-parallel kernels
// (kernels_cdf_par.hip) that lock-step decode needs for latency.
// HBM-bound in the limit: algorithmic bytes = vocab * 4 B of logits per token.
#include "cz_cdf_common.h"


__global__ void cdf_pairs_serial_kernel(const float* __restrict__ logits, int ld, const uint32_t* __restrict__ sym,
                                        uint32_t* __restrict__ c_lo, uint32_t* __restrict__ c_hi, int rows, int V, int mode) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    uint32_t lo, hi;
    serial_pairs_row(logits + (size_t)row * ld, V, mode, sym[row], &lo, &hi);
    c_lo[row] = lo; c_hi[row] = hi;
}
hipError_t cz_launch_cdf_pairs_serial(hipStream_t st, const float* logits, int ld, const uint32_t* sym, uint32_t* c_lo, uint32_t* c_hi,
                                      int rows, int vocab, int mode) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(cdf_pairs_serial_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, logits, ld, sym, c_lo, c_hi, rows, vocab, mode);
    return hipGetLastError();
}

__global__ void dec_init_kernel(cz_dec_state* st, int n) {                  // ArithmeticDecoder::new :419-436
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cz_dec_state d = st[i];
    d.low = 0; d.high = AC_FULL; d.code = 0; d.byte_pos = 0; d.bit_pos = 0;
    for (int k = 0; k < 32; ++k) d.code = (d.code << 1) | dec_get_bit(d);
    st[i] = d;
}
// more payload bytes have arrived (cz_decode_feed_batch): new length per stream, and whether the payloads are now complete
__global__ void dec_feed_kernel(cz_dec_state* st, int n, const unsigned long long* __restrict__ new_len, int final_input) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st[i].in_len = new_len[i];
    st[i].final_input = final_input ? 1u : 0u;
}
hipError_t cz_launch_dec_feed(hipStream_t s, cz_dec_state* st, int n, const unsigned long long* new_len_dev, int final_input) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(dec_feed_kernel, dim3((n + 63) / 64), dim3(64), 0, s, st, n, new_len_dev, final_input);
    return hipGetLastError();
}
hipError_t cz_launch_dec_init(hipStream_t s, cz_dec_state* st, int n) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(dec_init_kernel, dim3((n + 63) / 64), dim3(64), 0, s, st, n);
    return hipGetLastError();
}

__global__ void decode_symbol_serial_kernel(const float* __restrict__ logits, int ld, cz_dec_state* __restrict__ st,
                                            uint32_t* __restrict__ sym_out, int nstreams, int V, int mode, unsigned long long i,
                                            const unsigned long long* __restrict__ step_dev) {
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    if (step_dev) i = *step_dev;
    cz_dec_state d = st[sidx];
    if (!d.active || i >= d.token_count) { if (d.active) { d.active = 0; st[sidx] = d; } return; }
    uint32_t s = 0;
    if (!serial_decode_row(logits + (size_t)sidx * ld, V, mode, d, &s)) { d.active = 0; st[sidx] = d; sym_out[sidx] = 0; return; }
    d.out_ids[i + 1] = s;
    sym_out[sidx] = s;
    st[sidx] = d;
}
hipError_t cz_launch_decode_symbol_serial(hipStream_t s, const float* logits, int ld, cz_dec_state* st, uint32_t* sym_out, int nstreams,
                                          int vocab, int mode, unsigned long long i, const unsigned long long* step_dev) {
    if (nstreams <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_symbol_serial_kernel, dim3((nstreams + 63) / 64), dim3(64), 0, s, logits, ld, st, sym_out, nstreams, vocab, mode, i, step_dev);
    return hipGetLastError();
}

/* internal helpers shared by the oracle's translation units (test infrastructure only) */
#ifndef CZ_ORACLE_INTERNAL_H
#define CZ_ORACLE_INTERNAL_H
#include "cz_oracle.h"
#include <string.h>
static inline float czo_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t czo_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float czo_bf16_to_f32(uint16_t b) { return czo_bits_f32((uint32_t)b << 16); }
typedef struct { int N, K; float* p; } czo_packed_w;      /* f32 panels of 16 output columns: P[n/16][k][16] */
void czo_pack_weights(czo_packed_w* pw, const uint16_t* w_bf16, int N, int K);
/* Y[m][n] = cz_dot(X[m][:], W[n][:]) in the canonical chunk-64 fmaf order */
void czo_gemm_canon(const float* X, int ldx, const czo_packed_w* W, float* Y, int ldy, int M);
uint32_t czo_fnv1a32(const char* s);
#endif

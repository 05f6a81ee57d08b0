/*
 * cz_oracle_blake3.c — CPU ORACLE: BLAKE3 (unkeyed hash), truncated to 16 bytes as the container uses it
 * (reference src/main.rs:901-939: blake3_bytes_bin16 / blake3_file_bin16 / blake3_files_bin16 via the
 * third-party `blake3 = "1.5"` crate, Cargo.toml:18 — not in tree; restated from the published BLAKE3 spec).
 * Pinned by the orig_hash16 fields of the reference's committed .canz files against the input files they
 * were produced from (tests/test_oracle_container.py).  Test infrastructure only.
 */
#include "cz_oracle_internal.h"
#include <stdlib.h>

static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define G(a, b, c, d, mx, my) \
    do { s[a] = s[a] + s[b] + (mx); s[d] = rotr(s[d] ^ s[a], 16); s[c] = s[c] + s[d]; s[b] = rotr(s[b] ^ s[c], 12); \
         s[a] = s[a] + s[b] + (my); s[d] = rotr(s[d] ^ s[a], 8);  s[c] = s[c] + s[d]; s[b] = rotr(s[b] ^ s[c], 7); } while (0)

static void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t s[16], m[16], t[16];
    for (int i = 0; i < 8; ++i) s[i] = cv[i];
    for (int i = 0; i < 4; ++i) s[8 + i] = IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
    for (int i = 0; i < 16; ++i) m[i] = block[i];
    for (int r = 0; r < 7; ++r) {
        G(0, 4, 8, 12, m[0], m[1]); G(1, 5, 9, 13, m[2], m[3]); G(2, 6, 10, 14, m[4], m[5]); G(3, 7, 11, 15, m[6], m[7]);
        G(0, 5, 10, 15, m[8], m[9]); G(1, 6, 11, 12, m[10], m[11]); G(2, 7, 8, 13, m[12], m[13]); G(3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; ++i) t[i] = m[PERM[i]];
        for (int i = 0; i < 16; ++i) m[i] = t[i];
    }
    for (int i = 0; i < 8; ++i) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
static void load_block(const uint8_t* p, size_t n, uint32_t w[16]) {
    uint8_t b[64]; memset(b, 0, 64); memcpy(b, p, n);
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
}
/* an "output" = the inputs of a final compression, so the ROOT flag can be added late */
typedef struct { uint32_t cv[8], block[16]; uint64_t counter; uint32_t block_len, flags; } b3_output;

static void chunk_output(const uint8_t* p, size_t n, uint64_t chunk_counter, b3_output* o) {
    uint32_t cv[8], st[16]; memcpy(cv, IV, sizeof cv);
    size_t nblocks = n == 0 ? 1 : (n + 63) / 64;
    for (size_t b = 0; b < nblocks; ++b) {
        size_t off = b * 64, len = n - off < 64 ? n - off : 64;
        uint32_t w[16]; load_block(p + off, len, w);
        uint32_t fl = (b == 0 ? CHUNK_START : 0) | (b + 1 == nblocks ? CHUNK_END : 0);
        if (b + 1 == nblocks) { memcpy(o->cv, cv, sizeof cv); memcpy(o->block, w, sizeof w); o->counter = chunk_counter; o->block_len = (uint32_t)len; o->flags = fl; return; }
        compress(cv, w, chunk_counter, 64, fl, st);
        memcpy(cv, st, sizeof cv);
    }
}
static void output_cv(const b3_output* o, uint32_t cv[8]) { uint32_t st[16]; compress(o->cv, o->block, o->counter, o->block_len, o->flags, st); memcpy(cv, st, 32); }
static void parent_output(const uint32_t l[8], const uint32_t r[8], b3_output* o) {
    memcpy(o->cv, IV, 32); memcpy(o->block, l, 32); memcpy(o->block + 8, r, 32); o->counter = 0; o->block_len = 64; o->flags = PARENT;
}

/* incremental state so that several byte strings can be fed (blake3_files_bin16 hashes name then content) */
struct czo_blake3 { uint32_t stack[54][8]; int sp; uint64_t chunks; uint8_t buf[1024]; size_t buf_len; };
typedef struct czo_blake3 czo_blake3;
czo_blake3* czo_blake3_new(void) { return (czo_blake3*)calloc(1, sizeof(czo_blake3)); }
static void push_chunk_cv(czo_blake3* h, uint32_t cv[8]) {
    uint64_t total = h->chunks;               /* chunks completed including this one */
    while ((total & 1) == 0) { b3_output o; parent_output(h->stack[--h->sp], cv, &o); output_cv(&o, cv); total >>= 1; }
    memcpy(h->stack[h->sp++], cv, 32);
}
void czo_blake3_update(czo_blake3* h, const uint8_t* p, size_t n) {
    while (n > 0) {
        if (h->buf_len == 1024) {             /* more input follows: the buffered chunk is not the last one */
            b3_output o; uint32_t cv[8];
            chunk_output(h->buf, 1024, h->chunks, &o); output_cv(&o, cv);
            h->chunks += 1; push_chunk_cv(h, cv); h->buf_len = 0;
        }
        size_t take = 1024 - h->buf_len < n ? 1024 - h->buf_len : n;
        memcpy(h->buf + h->buf_len, p, take); h->buf_len += take; p += take; n -= take;
    }
}
void czo_blake3_final16(czo_blake3* h, uint8_t out[16]) {
    b3_output o; chunk_output(h->buf, h->buf_len, h->chunks, &o);
    int sp = h->sp;
    while (sp > 0) { uint32_t cv[8]; output_cv(&o, cv); b3_output po; parent_output(h->stack[--sp], cv, &po); o = po; }
    uint32_t st[16]; compress(o.cv, o.block, 0, o.block_len, o.flags | ROOT, st);
    for (int i = 0; i < 4; ++i) { out[4 * i] = (uint8_t)st[i]; out[4 * i + 1] = (uint8_t)(st[i] >> 8); out[4 * i + 2] = (uint8_t)(st[i] >> 16); out[4 * i + 3] = (uint8_t)(st[i] >> 24); }
}
void czo_blake3_free(czo_blake3* h) { free(h); }
void czo_blake3_bytes_bin16(const uint8_t* p, size_t n, uint8_t out[16]) {    /* src/main.rs:901-906 */
    czo_blake3* h = czo_blake3_new(); czo_blake3_update(h, p, n); czo_blake3_final16(h, out); czo_blake3_free(h);
}

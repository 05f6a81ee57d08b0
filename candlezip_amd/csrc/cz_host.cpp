// cz_host.cpp — host-side (non-kernel) parts of libcandlezip_hip: error text, JSON, safetensors, BLAKE3-128,
// arithmetic encoder, .canz header.  Each C-ABI function cites the reference function it stands in for.
#include "cz_host.h"
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void cz_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char* cz_last_error(void) { return g_err; }
extern "C" void cz_free(void* p) { free(p); }

// ------------------------------------------------------------------ JSON
const JsonValue* JsonValue::get(const char* key) const {
    if (kind != Obj) return nullptr;
    for (auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
}
namespace {
struct JP {
    const char* p; const char* e; std::string err;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    static void put_utf8(std::string& s, uint32_t c) {
        if (c < 0x80) s += (char)c;
        else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
        else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    }
    bool hex4(uint32_t& v) {
        if (e - p < 4) return fail("bad \\u");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++; v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10; else return fail("bad hex");
        }
        return true;
    }
    bool str(std::string& s) {
        if (p >= e || *p != '"') return fail("expected string");
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                ++p; if (p >= e) return fail("bad escape");
                char c = *p++;
                switch (c) {
                    case '"': s += '"'; break; case '\\': s += '\\'; break; case '/': s += '/'; break;
                    case 'b': s += '\b'; break; case 'f': s += '\f'; break; case 'n': s += '\n'; break;
                    case 'r': s += '\r'; break; case 't': s += '\t'; break;
                    case 'u': {
                        uint32_t v; if (!hex4(v)) return false;
                        if (v >= 0xD800 && v < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            p += 2; uint32_t lo; if (!hex4(lo)) return false;
                            v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        put_utf8(s, v); break;
                    }
                    default: return fail("bad escape");
                }
            } else s += *p++;
        }
        if (p >= e) return fail("unterminated string");
        ++p; return true;
    }
    bool val(JsonValue& v, int depth) {
        if (depth > 64) return fail("too deep");
        ws(); if (p >= e) return fail("eof");
        if (*p == '{') {
            v.kind = JsonValue::Obj; ++p; ws();
            if (p < e && *p == '}') { ++p; return true; }
            for (;;) {
                ws(); std::string k; if (!str(k)) return false;
                ws(); if (p >= e || *p != ':') return fail("expected :"); ++p;
                v.obj.emplace_back(std::move(k), JsonValue());
                if (!val(v.obj.back().second, depth + 1)) return false;
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; return true; }
                return fail("expected , or }");
            }
        }
        if (*p == '[') {
            v.kind = JsonValue::Arr; ++p; ws();
            if (p < e && *p == ']') { ++p; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!val(v.arr.back(), depth + 1)) return false;
                ws(); if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; return true; }
                return fail("expected , or ]");
            }
        }
        if (*p == '"') { v.kind = JsonValue::Str; return str(v.str); }
        if (e - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JsonValue::Bool; v.b = true; p += 4; return true; }
        if (e - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JsonValue::Bool; v.b = false; p += 5; return true; }
        if (e - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JsonValue::Null; p += 4; return true; }
        char* end = nullptr; std::string tmp(p, (size_t)((e - p) < 64 ? (e - p) : 64));
        double d = strtod(tmp.c_str(), &end);
        if (end == tmp.c_str()) return fail("bad value");
        v.kind = JsonValue::Num; v.num = d; p += end - tmp.c_str(); return true;
    }
};
}  // namespace
bool cz_json_parse(const char* text, size_t len, JsonValue* out, std::string* err) {
    JP jp{text, text + len, {}};
    bool ok = jp.val(*out, 0);
    if (!ok && err) *err = jp.err;
    return ok;
}

// ------------------------------------------------------------------ safetensors
SafeTensorsFile::~SafeTensorsFile() {
    if (map) munmap(map, map_len);
    if (fd >= 0) close(fd);
}
bool cz_safetensors_open(const char* path, SafeTensorsFile* out) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { cz_set_error("failed reading %s", path); return false; }
    struct stat st; if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); cz_set_error("bad safetensors file %s", path); return false; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); cz_set_error("mmap failed for %s", path); return false; }
    out->map = m; out->map_len = (size_t)st.st_size; out->fd = fd;
    const uint8_t* b = (const uint8_t*)m;
    uint64_t hl = 0; for (int i = 0; i < 8; ++i) hl |= (uint64_t)b[i] << (8 * i);
    if (hl > out->map_len - 8) { cz_set_error("safetensors header length out of range in %s", path); return false; }
    JsonValue root; std::string err;
    if (!cz_json_parse((const char*)b + 8, (size_t)hl, &root, &err) || root.kind != JsonValue::Obj) {
        cz_set_error("safetensors header of %s is not valid JSON: %s", path, err.c_str()); return false;
    }
    const uint8_t* data0 = b + 8 + hl; size_t data_len = out->map_len - 8 - (size_t)hl;
    for (auto& kv : root.obj) {
        if (kv.first == "__metadata__") continue;
        const JsonValue *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
        if (!dt || !sh || !off || off->arr.size() != 2) { cz_set_error("malformed tensor entry %s", kv.first.c_str()); return false; }
        SafeTensor t; t.dtype = dt->str;
        for (auto& d : sh->arr) t.shape.push_back((int64_t)d.num);
        size_t s = (size_t)off->arr[0].num, e = (size_t)off->arr[1].num;
        if (e < s || e > data_len) { cz_set_error("tensor %s out of file bounds", kv.first.c_str()); return false; }
        t.data = data0 + s; t.nbytes = e - s;
        out->tensors[kv.first] = t;
    }
    return true;
}

// ------------------------------------------------------------------ BLAKE3 (src/main.rs:901-939 use blake3 = "1.5")
namespace {
const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline void b3_g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr32(s[d] ^ s[a], 16); s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr32(s[d] ^ s[a], 8);  s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 7);
}
void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t blen, uint32_t flags, uint32_t out[16]) {
    uint32_t s[16], m[16], t[16];
    for (int i = 0; i < 8; ++i) s[i] = cv[i];
    for (int i = 0; i < 4; ++i) s[8 + i] = B3_IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = blen; s[15] = flags;
    memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; ++r) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]); b3_g(s, 1, 5, 9, 13, m[2], m[3]); b3_g(s, 2, 6, 10, 14, m[4], m[5]); b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]); b3_g(s, 1, 6, 11, 12, m[10], m[11]); b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; ++i) t[i] = m[B3_PERM[i]];
        memcpy(m, t, sizeof m);
    }
    for (int i = 0; i < 8; ++i) { out[i] = s[i] ^ s[i + 8]; out[i + 8] = s[i + 8] ^ cv[i]; }
}
struct B3Out { uint32_t cv[8], block[16]; uint64_t counter; uint32_t blen, flags; };
void b3_words(const uint8_t* p, size_t n, uint32_t w[16]) {
    uint8_t b[64]; memset(b, 0, 64); memcpy(b, p, n);
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
}
void b3_chunk(const uint8_t* p, size_t n, uint64_t ctr, B3Out* o) {
    uint32_t cv[8], st[16]; memcpy(cv, B3_IV, sizeof cv);
    size_t nb = n == 0 ? 1 : (n + 63) / 64;
    for (size_t b = 0; b < nb; ++b) {
        size_t off = b * 64, len = n - off < 64 ? n - off : 64;
        uint32_t w[16]; b3_words(p + off, len, w);
        uint32_t fl = (b == 0 ? B3_CHUNK_START : 0) | (b + 1 == nb ? B3_CHUNK_END : 0);
        if (b + 1 == nb) { memcpy(o->cv, cv, 32); memcpy(o->block, w, 64); o->counter = ctr; o->blen = (uint32_t)len; o->flags = fl; return; }
        b3_compress(cv, w, ctr, 64, fl, st); memcpy(cv, st, 32);
    }
}
void b3_out_cv(const B3Out* o, uint32_t cv[8]) { uint32_t st[16]; b3_compress(o->cv, o->block, o->counter, o->blen, o->flags, st); memcpy(cv, st, 32); }
void b3_parent(const uint32_t l[8], const uint32_t r[8], B3Out* o) {
    memcpy(o->cv, B3_IV, 32); memcpy(o->block, l, 32); memcpy(o->block + 8, r, 32); o->counter = 0; o->blen = 64; o->flags = B3_PARENT;
}
}  // namespace
void Blake3::update(const uint8_t* p, size_t n) {
    while (n > 0) {
        if (buf_len == 1024) {
            B3Out o; uint32_t cv[8]; b3_chunk(buf, 1024, chunks, &o); b3_out_cv(&o, cv);
            chunks += 1;
            uint64_t total = chunks;
            while ((total & 1) == 0) { B3Out po; b3_parent(stack[--sp], cv, &po); b3_out_cv(&po, cv); total >>= 1; }
            memcpy(stack[sp++], cv, 32);
            buf_len = 0;
        }
        size_t take = 1024 - buf_len < n ? 1024 - buf_len : n;
        memcpy(buf + buf_len, p, take); buf_len += take; p += take; n -= take;
    }
}
void Blake3::final16(uint8_t out[16]) {
    B3Out o; b3_chunk(buf, buf_len, chunks, &o);
    int s = sp;
    while (s > 0) { uint32_t cv[8]; b3_out_cv(&o, cv); B3Out po; b3_parent(stack[--s], cv, &po); o = po; }
    uint32_t st[16]; b3_compress(o.cv, o.block, 0, o.blen, o.flags | B3_ROOT, st);
    for (int i = 0; i < 4; ++i) { out[4 * i] = (uint8_t)st[i]; out[4 * i + 1] = (uint8_t)(st[i] >> 8); out[4 * i + 2] = (uint8_t)(st[i] >> 16); out[4 * i + 3] = (uint8_t)(st[i] >> 24); }
}
extern "C" void cz_blake3_bytes_bin16(const uint8_t* p, size_t n, uint8_t out[16]) {       // src/main.rs:901-906
    Blake3 h; h.update(p, n); h.final16(out);
}
static bool hash_file_into(Blake3& h, const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { cz_set_error("cannot open %s", path); return false; }
    std::vector<uint8_t> buf(1 << 16);
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) h.update(buf.data(), n);
    fclose(f); return true;
}
extern "C" cz_status cz_blake3_file_bin16(const char* path, uint8_t out[16]) {              // src/main.rs:908-921
    Blake3 h; if (!hash_file_into(h, path)) return CZ_ERR_IO; h.final16(out); return CZ_OK;
}
extern "C" cz_status cz_blake3_files_bin16(const char* const* paths, size_t n, uint8_t out[16]) {   // src/main.rs:922-939
    Blake3 h;
    for (size_t i = 0; i < n; ++i) {
        const char* base = strrchr(paths[i], '/'); base = base ? base + 1 : paths[i];
        h.update((const uint8_t*)base, strlen(base));                                      // the file NAME is hashed first
        if (!hash_file_into(h, paths[i])) return CZ_ERR_IO;
    }
    h.final16(out); return CZ_OK;
}

// ------------------------------------------------------------------ arithmetic encoder (src/main.rs:261-318, 353-404)
static const uint64_t AC_FULL = 0xFFFFFFFFull, AC_HALF = 0x80000000ull, AC_QUARTER = 0x40000000ull;
void AcEncoder::put_bit_internal(uint8_t bit) {                                            // :298-308
    bit_buffer = (uint8_t)((bit_buffer << 1) | (bit & 1)); bit_count += 1;
    if (bit_count == 8) { out.push_back(bit_buffer); bit_buffer = 0; bit_count = 0; }
}
void AcEncoder::put_bit(uint8_t bit) {                                                     // :310-317
    put_bit_internal(bit);
    while (carry_run > 0) { put_bit_internal((uint8_t)((~bit) & 1)); carry_run -= 1; }
}
void AcEncoder::encode_counts(uint64_t c_lo, uint64_t c_hi, uint64_t total) {              // :353-385
    if (c_lo == c_hi) zero_width += 1;
    u128 range = (u128)(uint64_t)(high - low + 1);
    u128 nl = (u128)low + (range * (u128)c_lo) / (u128)total;
    u128 nh = (u128)low + (range * (u128)c_hi) / (u128)total - 1;                          // u128 wrap when 0, as a release build does
    low = (uint64_t)(nl & (u128)AC_FULL); high = (uint64_t)(nh & (u128)AC_FULL);
    for (;;) {
        if (high < AC_HALF) put_bit(0);
        else if (low >= AC_HALF) { put_bit(1); low -= AC_HALF; high -= AC_HALF; }
        else if (low >= AC_QUARTER && high < AC_QUARTER * 3) { carry_run += 1; low -= AC_QUARTER; high -= AC_QUARTER; }
        else break;
        low = (low << 1) & AC_FULL; high = ((high << 1) & AC_FULL) | 1;
    }
}
void AcEncoder::finish() {                                                                 // :387-399
    carry_run += 1;
    put_bit(low < AC_QUARTER ? 0 : 1);
    if (bit_count > 0) { int rem = 8 - bit_count; for (int i = 0; i < rem; ++i) put_bit_internal(0); }
}
extern "C" cz_status cz_ac_encode_pairs(const uint32_t* c_lo, const uint32_t* c_hi, size_t n, uint8_t** payload, size_t* len,
                                        uint64_t* zero_width) {
    if ((!c_lo || !c_hi) && n) { cz_set_error("cz_ac_encode_pairs: null interval arrays"); return CZ_ERR_ARG; }
    if (!payload || !len) { cz_set_error("cz_ac_encode_pairs: null output"); return CZ_ERR_ARG; }
    AcEncoder e;
    for (size_t i = 0; i < n; ++i) e.encode_counts(c_lo[i], c_hi[i], 1ull << 30);
    e.finish();
    *payload = (uint8_t*)malloc(e.out.size() ? e.out.size() : 1);
    memcpy(*payload, e.out.data(), e.out.size());
    *len = e.out.size();
    if (zero_width) *zero_width = e.zero_width;
    return CZ_OK;
}

// the same encoder fed in pieces: bytes come out as soon as they are final (a decoder elsewhere may already consume them)
struct cz_ac_encoder { AcEncoder e; size_t emitted = 0; bool finished = false; };
extern "C" cz_status cz_ac_encoder_new(cz_ac_encoder** out) {
    if (!out) { cz_set_error("cz_ac_encoder_new: null argument"); return CZ_ERR_ARG; }
    *out = new cz_ac_encoder();
    return CZ_OK;
}
extern "C" void cz_ac_encoder_free(cz_ac_encoder* enc) { delete enc; }
extern "C" cz_status cz_ac_encoder_push(cz_ac_encoder* enc, const uint32_t* c_lo, const uint32_t* c_hi, size_t n, const uint8_t** new_bytes, size_t* new_len) {
    if (!enc || ((!c_lo || !c_hi) && n) || !new_bytes || !new_len) { cz_set_error("cz_ac_encoder_push: null argument"); return CZ_ERR_ARG; }
    if (enc->finished) { cz_set_error("cz_ac_encoder_push: the encoder was finished"); return CZ_ERR_STATE; }
    for (size_t i = 0; i < n; ++i) enc->e.encode_counts(c_lo[i], c_hi[i], 1ull << 30);
    *new_bytes = enc->e.out.data() + enc->emitted; *new_len = enc->e.out.size() - enc->emitted;
    enc->emitted = enc->e.out.size();
    return CZ_OK;
}
extern "C" cz_status cz_ac_encoder_finish(cz_ac_encoder* enc, const uint8_t** new_bytes, size_t* new_len, size_t* total_len, uint64_t* zero_width) {
    if (!enc || !new_bytes || !new_len) { cz_set_error("cz_ac_encoder_finish: null argument"); return CZ_ERR_ARG; }
    if (!enc->finished) { enc->e.finish(); enc->finished = true; }
    *new_bytes = enc->e.out.data() + enc->emitted; *new_len = enc->e.out.size() - enc->emitted;
    enc->emitted = enc->e.out.size();
    if (total_len) *total_len = enc->e.out.size();
    if (zero_width) *zero_width = enc->e.zero_width;
    return CZ_OK;
}

// ------------------------------------------------------------------ container header v2 (src/main.rs:551-646)
static size_t put_var(uint64_t v, uint8_t* o) { size_t n = 0; while (v >= 0x80) { o[n++] = (uint8_t)((v & 0x7F) | 0x80); v >>= 7; } o[n++] = (uint8_t)v; return n; }
static size_t get_var(const uint8_t* in, size_t len, uint64_t* v) {
    uint32_t shift = 0; uint64_t out = 0; size_t n = 0;
    for (;;) {
        if (n >= len) return 0;
        uint8_t b = in[n++]; out |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
        shift += 7; if (shift > 63) return 0;                                              // "varint too long"
    }
    *v = out; return n;
}
static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

extern "C" size_t cz_write_header_v2(const cz_header* h, const uint8_t* repr, uint8_t* out, size_t cap) {
    if (cap < 98 + (size_t)h->model_file_repr_len) return 0;
    size_t n = 0;
    put32(out, 0x5a505447u); n = 4;                                                        // MAGIC "GPTZ" :227
    out[n++] = 2; out[n++] = 0;                                                            // VERSION :228
    put32(out + n, h->bos_token_id); n += 4;
    n += put_var(h->token_count, out + n);
    n += put_var(h->orig_len_bytes, out + n);
    memcpy(out + n, h->model_hash16, 16); n += 16;
    memcpy(out + n, h->tokenizer_hash16, 16); n += 16;
    memcpy(out + n, h->orig_hash16, 16); n += 16;
    put32(out + n, h->reserved_flags); n += 4;
    put32(out + n, h->context_window); n += 4;
    put32(out + n, h->vocab_size); n += 4;
    put32(out + n, h->model_file_repr_len); n += 4;
    put32(out + n, h->reprime_interval); n += 4;
    if (h->model_file_repr_len) memcpy(out + n, repr, h->model_file_repr_len);
    return n + h->model_file_repr_len;
}
extern "C" cz_status cz_read_header_v2(const uint8_t* buf, size_t len, cz_header* h, size_t* repr_off, size_t* consumed) {
    size_t n, k;
    if (len < 10) { cz_set_error("container truncated"); return CZ_ERR_FORMAT; }
    if (get32(buf) != 0x5a505447u) { cz_set_error("bad magic"); return CZ_ERR_FORMAT; }     // :616
    if (buf[4] != 2 || buf[5] != 0) { cz_set_error("bad version"); return CZ_ERR_FORMAT; }  // :618
    n = 6; h->bos_token_id = get32(buf + n); n += 4;
    if (!(k = get_var(buf + n, len - n, &h->token_count))) { cz_set_error("bad varint (token_count)"); return CZ_ERR_FORMAT; }
    n += k;
    if (!(k = get_var(buf + n, len - n, &h->orig_len_bytes))) { cz_set_error("bad varint (orig_len)"); return CZ_ERR_FORMAT; }
    n += k;
    if (len - n < 68) { cz_set_error("container truncated"); return CZ_ERR_FORMAT; }
    memcpy(h->model_hash16, buf + n, 16); n += 16;
    memcpy(h->tokenizer_hash16, buf + n, 16); n += 16;
    memcpy(h->orig_hash16, buf + n, 16); n += 16;
    h->reserved_flags = get32(buf + n); n += 4;
    h->context_window = get32(buf + n); n += 4;
    h->vocab_size = get32(buf + n); n += 4;
    h->model_file_repr_len = get32(buf + n); n += 4;
    h->reprime_interval = get32(buf + n); n += 4;
    if (len - n < h->model_file_repr_len) { cz_set_error("container truncated"); return CZ_ERR_FORMAT; }
    if (repr_off) *repr_off = n;
    if (consumed) *consumed = n + h->model_file_repr_len;
    return CZ_OK;
}
extern "C" uint32_t cz_flags_pack(int agent_used, int agent_mock, int gates_present, uint32_t agent_chunk) {   // :247-255
    uint32_t f = 0;
    if (agent_used) f |= 1u; if (agent_mock) f |= 2u; if (gates_present) f |= 4u;
    return f | ((agent_chunk & 0xFFFFu) << 16);
}
extern "C" cz_status cz_skip_gate_section(const uint8_t* buf, size_t len, size_t* consumed) {   // src/main.rs:2469-2484
    if (len < 4) { cz_set_error("invalid gating magic"); return CZ_ERR_FORMAT; }
    uint64_t cnt; size_t k;
    if (!memcmp(buf, "AGT2", 4)) {
        if (!(k = get_var(buf + 4, len - 4, &cnt)) || len - 4 - k < cnt) { cz_set_error("gate section truncated"); return CZ_ERR_FORMAT; }
        *consumed = 4 + k + (size_t)cnt; return CZ_OK;
    }
    if (!memcmp(buf, "AGTB", 4)) {
        if (!(k = get_var(buf + 4, len - 4, &cnt))) { cz_set_error("gate section truncated"); return CZ_ERR_FORMAT; }
        size_t nb = (size_t)((cnt + 7) / 8);
        if (len - 4 - k < nb) { cz_set_error("gate section truncated"); return CZ_ERR_FORMAT; }
        *consumed = 4 + k + nb; return CZ_OK;
    }
    cz_set_error("invalid gating magic"); return CZ_ERR_FORMAT;
}

// ------------------------------------------------------------------ RWKV "World" tokenizer
// candle_rwkv7/src/models/rwkv7.rs:546-610 (Tokenizer::new / encode_bytes / decode_bytes) and the symbol planning of
// src/main.rs:833-895 (plan_rwkv_symbols / rwkv_detok_with_literals).
#include <unordered_map>
struct cz_rwkv_tokenizer {
    std::vector<std::vector<std::string>> table;          // [s0*256 + s1] tokens of length >= 2, descending id (rwkv7.rs:562-570)
    std::vector<uint8_t> good;                            // [s0*256 + s1]
    std::unordered_map<std::string, uint32_t> token2idx;
    std::unordered_map<uint32_t, std::string> idx2token;
};
extern "C" cz_status cz_rwkv_tokenizer_open(const char* path, cz_rwkv_tokenizer** out) {
    if (!path || !out) { cz_set_error("cz_rwkv_tokenizer_open: null argument"); return CZ_ERR_ARG; }
    FILE* f = fopen(path, "rb");
    if (!f) { cz_set_error("cannot open %s", path); return CZ_ERR_IO; }
    std::string txt; char buf[1 << 14]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, n);
    fclose(f);
    JsonValue root; std::string err;
    if (!cz_json_parse(txt.data(), txt.size(), &root, &err) || root.kind != JsonValue::Obj) { cz_set_error("rwkv vocab json: %s", err.c_str()); return CZ_ERR_IO; }
    cz_rwkv_tokenizer* t = new cz_rwkv_tokenizer();
    t->table.resize(65536); t->good.assign(65536, 0);
    uint32_t max_idx = 0;
    for (auto& kv : root.obj) {                                                     // HashMap<String, u32>: the last duplicate key wins
        if (kv.second.kind != JsonValue::Num) { delete t; cz_set_error("rwkv vocab json: value of \"%s\" is not a number", kv.first.c_str()); return CZ_ERR_IO; }
        t->token2idx[kv.first] = (uint32_t)kv.second.num;
    }
    // idx2token from the final map (rwkv7.rs:557).  Two keys with one id: the reference keeps whichever its HashMap iteration meets
    // last (unspecified); here the last one in file order, deterministically.
    for (auto& kv : root.obj) {
        const uint32_t id = (uint32_t)kv.second.num;
        if (t->token2idx[kv.first] != id) continue;
        t->idx2token[id] = kv.first;
        if (id > max_idx) max_idx = id;
    }
    for (uint64_t idx = (uint64_t)max_idx + 1; idx-- > 0;) {                       // (0..1+max_idx).rev()
        auto it = t->idx2token.find((uint32_t)idx);
        if (it == t->idx2token.end()) continue;
        const std::string& s = it->second;
        if (s.size() >= 2) {
            size_t key = (size_t)(uint8_t)s[0] * 256 + (uint8_t)s[1];
            t->table[key].push_back(s); t->good[key] = 1;
        }
    }
    *out = t;
    return CZ_OK;
}
extern "C" void cz_rwkv_tokenizer_close(cz_rwkv_tokenizer* t) { delete t; }
static bool rwkv_encode_bytes(const cz_rwkv_tokenizer* t, const uint8_t* b, size_t n, std::vector<uint32_t>* ids) {   // rwkv7.rs:584-607
    size_t i = 0;
    while (i < n) {
        std::string s(1, (char)b[i]);
        if (i + 1 < n) {
            size_t key = (size_t)b[i] * 256 + b[i + 1];
            if (t->good[key])
                for (auto& cand : t->table[key])
                    if (cand.size() <= n - i && !memcmp(cand.data(), b + i, cand.size())) { s = cand; break; }
        }
        i += s.size();
        auto it = t->token2idx.find(s);
        if (it == t->token2idx.end()) return false;                               // "Tokenizer vocabulary missing byte sequence"
        ids->push_back(it->second);
    }
    return true;
}
extern "C" cz_status cz_rwkv_plan_symbols(const cz_rwkv_tokenizer* t, const uint8_t* bytes, size_t n_bytes, uint32_t vocab,
                                          uint32_t* ids_out, size_t ids_cap, size_t* n_ids) {       // src/main.rs:833-870
    if (!t || (!bytes && n_bytes) || (!ids_out && ids_cap) || !n_ids) { cz_set_error("cz_rwkv_plan_symbols: null argument"); return CZ_ERR_ARG; }
    std::vector<uint32_t> ids;
    if (rwkv_encode_bytes(t, bytes, n_bytes, &ids)) {
        size_t covered = 0;
        for (uint32_t id : ids) { auto it = t->idx2token.find(id); if (it != t->idx2token.end()) covered += it->second.size(); }
        for (size_t i = covered; i < n_bytes; ++i) ids.push_back(vocab + bytes[i]);
    } else {
        ids.clear();
        for (size_t i = 0; i < n_bytes; ++i) ids.push_back(vocab + bytes[i]);      // complete fallback: literals
    }
    *n_ids = ids.size();
    if (ids.size() > ids_cap) { cz_set_error("cz_rwkv_plan_symbols: %zu symbols do not fit ids_cap = %zu", ids.size(), ids_cap); return CZ_ERR_ARG; }
    if (!ids.empty()) memcpy(ids_out, ids.data(), ids.size() * 4);
    return CZ_OK;
}
extern "C" cz_status cz_rwkv_detok(const cz_rwkv_tokenizer* t, const uint32_t* ids, size_t n, uint32_t vocab, uint8_t** out, size_t* out_len) {
    if (!t || (!ids && n) || !out || !out_len) { cz_set_error("cz_rwkv_detok: null argument"); return CZ_ERR_ARG; }   // src/main.rs:872-895
    std::string o;
    for (size_t i = 0; i < n; ++i) {
        if (ids[i] < vocab) { auto it = t->idx2token.find(ids[i]); if (it != t->idx2token.end()) o += it->second; }
        else o += (char)(uint8_t)(ids[i] - vocab);
    }
    *out = (uint8_t*)malloc(o.size() ? o.size() : 1);
    memcpy(*out, o.data(), o.size());
    *out_len = o.size();
    return CZ_OK;
}

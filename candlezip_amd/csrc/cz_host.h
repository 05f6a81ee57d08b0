// cz_host.h — host-side pieces of libcandlezip_hip that are not kernels: error text, JSON, safetensors, BLAKE3,
// arithmetic encoder, container.  (The reference host code is Rust; Rust is not available in this toolchain, so the
// host side above the C ABI is C++ mirroring the reference's interfaces — see INTEGRATION.md.)
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/candlezip.h"

void cz_set_error(const char* fmt, ...);

// ---- minimal JSON (config.json, safetensors header, rwkv vocab) ----
struct JsonValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;                                    // raw bytes after unescaping (UTF-8)
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj; // insertion order kept
    const JsonValue* get(const char* key) const;
};
bool cz_json_parse(const char* text, size_t len, JsonValue* out, std::string* err);

// ---- safetensors (mmap'd) ----
struct SafeTensor { std::string dtype; std::vector<int64_t> shape; const uint8_t* data; size_t nbytes; };
struct SafeTensorsFile {
    void* map = nullptr; size_t map_len = 0; int fd = -1;
    std::map<std::string, SafeTensor> tensors;
    ~SafeTensorsFile();
};
bool cz_safetensors_open(const char* path, SafeTensorsFile* out);

// ---- BLAKE3 (unkeyed), incremental ----
struct Blake3 {
    uint32_t stack[54][8]; int sp = 0; uint64_t chunks = 0; uint8_t buf[1024]; size_t buf_len = 0;
    void update(const uint8_t* p, size_t n);
    void final16(uint8_t out[16]);
};

// ---- arithmetic encoder (src/main.rs:261-318, 353-404) ----
struct AcEncoder {
    uint64_t low = 0, high = 0xFFFFFFFFull, carry_run = 0;
    uint8_t bit_buffer = 0, bit_count = 0;
    std::vector<uint8_t> out;
    uint64_t zero_width = 0;
    void put_bit_internal(uint8_t bit);
    void put_bit(uint8_t bit);
    void encode_counts(uint64_t c_lo, uint64_t c_hi, uint64_t total);
    void finish();
};

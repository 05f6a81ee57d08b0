// cz_bpe.cpp — byte-level BPE tokenizer, in-process.
//
// Replaces what the reference gets from the HF `tokenizers` crate (Cargo.toml:25) for the SmolLM backend:
//   Tokenizer::from_file(tokenizer.json)                    src/main.rs:1812
//   tok.encode(String::from_utf8_lossy(&data), false)       src/main.rs:1852   (no special tokens added by a post-processor)
//   tokenizer.decode(&out_tokens[1..], true)                src/main.rs:2654   (skip special tokens)
// Supported tokenizer.json: model BPE (vocab + merges, ignore_merges), no normalizer, pre_tokenizer = ByteLevel(use_regex) optionally
// after Digits(individual_digits) inside a Sequence — the GPT-2 family layout SmolLM2 uses —, ByteLevel decoder, added tokens matched
// literally in the raw text.  Anything else is refused (CZ_ERR_UNSUPPORTED), never approximated.
// The published algorithm restated here: GPT-2 byte<->unicode alphabet; pre-tokenisation pattern
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
// (first alternative that matches wins); per piece, merges applied lowest rank first, leftmost first.
// Parity: pinned against the HF library itself on a tokenizer of this kind trained from the committed reference inputs
// (tests/golden/make_bpe_fixture.py); SmolLM2's own vocabulary is not available offline.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <memory>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>
#include "cz_host.h"
#include "cz_unicode_tables.h"

#define CZCHK(expr) do { cz_status _s = (expr); if (_s != CZ_OK) return _s; } while (0)

namespace {

bool in_ranges(const CzCpRange* r, int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (cp < r[mid].lo) hi = mid - 1; else if (cp > r[mid].hi) lo = mid + 1; else return true;
    }
    return false;
}
inline bool is_letter(uint32_t c) { return in_ranges(kLetter, kLetterCount, c); }
inline bool is_number(uint32_t c) { return in_ranges(kNumber, kNumberCount, c); }
inline bool is_white(uint32_t c) { return in_ranges(kWhite, kWhiteCount, c); }

void put_utf8(uint32_t cp, std::string* out) {
    if (cp < 0x80) out->push_back((char)cp);
    else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
    else { out->push_back((char)(0xF0 | (cp >> 18))); out->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
}
// String::from_utf8_lossy: every maximal invalid prefix of a sequence becomes one U+FFFD (Unicode "substitution of maximal subparts")
void decode_utf8_lossy(const uint8_t* s, size_t n, std::vector<uint32_t>* cps) {
    size_t i = 0;
    while (i < n) {
        const uint8_t b0 = s[i];
        if (b0 < 0x80) { cps->push_back(b0); ++i; continue; }
        int need = 0; uint32_t cp = 0; uint8_t lo = 0x80, hi = 0xBF;
        if (b0 >= 0xC2 && b0 <= 0xDF) { need = 1; cp = b0 & 0x1F; }
        else if (b0 >= 0xE0 && b0 <= 0xEF) { need = 2; cp = b0 & 0x0F; if (b0 == 0xE0) lo = 0xA0; if (b0 == 0xED) hi = 0x9F; }
        else if (b0 >= 0xF0 && b0 <= 0xF4) { need = 3; cp = b0 & 0x07; if (b0 == 0xF0) lo = 0x90; if (b0 == 0xF4) hi = 0x8F; }
        else { cps->push_back(0xFFFD); ++i; continue; }
        size_t j = i + 1; int got = 0;
        while (got < need && j < n) {
            const uint8_t b = s[j];
            const uint8_t l = got == 0 ? lo : 0x80, h = got == 0 ? hi : 0xBF;
            if (b < l || b > h) break;
            cp = (cp << 6) | (b & 0x3F); ++j; ++got;
        }
        if (got == need) cps->push_back(cp); else cps->push_back(0xFFFD);
        i = j;
    }
}

struct ByteAlphabet {
    uint32_t b2u[256]; std::unordered_map<uint32_t, uint8_t> u2b;
    ByteAlphabet() {                                              // GPT-2 bytes_to_unicode
        bool direct[256] = {false};
        for (int b = '!'; b <= '~'; ++b) direct[b] = true;
        for (int b = 0xA1; b <= 0xAC; ++b) direct[b] = true;
        for (int b = 0xAE; b <= 0xFF; ++b) direct[b] = true;
        uint32_t next = 256;
        for (int b = 0; b < 256; ++b) { b2u[b] = direct[b] ? (uint32_t)b : next++; u2b[b2u[b]] = (uint8_t)b; }
    }
};
const ByteAlphabet& alphabet() { static ByteAlphabet a; return a; }

struct PairHash { size_t operator()(const std::pair<uint32_t, uint32_t>& p) const { return ((size_t)p.first << 32) ^ p.second; } };

}  // namespace

struct cz_bpe_tokenizer {
    std::vector<std::string> id2bytes;                            // token id -> raw bytes (byte-level alphabet undone; added tokens verbatim)
    std::vector<uint8_t> is_special;
    int32_t byte_id[256];
    std::unordered_map<std::string, uint32_t> bytes2id;
    std::unordered_map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint32_t>, PairHash> merges;   // (a, b) -> (rank, merged id)
    struct Added { std::string content; uint32_t id; bool special; };
    std::vector<Added> added;
    bool digits = false, individual_digits = false, use_regex = true, ignore_merges = false;
    mutable std::unordered_map<std::string, std::vector<uint32_t>> cache;
};

static bool token_to_bytes(const std::string& tok, std::string* out) {          // undo the byte-level alphabet; false if a char is outside it
    std::vector<uint32_t> cps;
    decode_utf8_lossy((const uint8_t*)tok.data(), tok.size(), &cps);
    const auto& A = alphabet();
    out->clear();
    for (uint32_t c : cps) { auto it = A.u2b.find(c); if (it == A.u2b.end()) return false; out->push_back((char)it->second); }
    return true;
}

static cz_status parse_pretok(const JsonValue* pt, cz_bpe_tokenizer* t, bool* seen_bytelevel) {
    if (!pt || pt->kind != JsonValue::Obj) { cz_set_error("tokenizer.json: pre_tokenizer missing (only byte-level BPE tokenizers are supported)"); return CZ_ERR_UNSUPPORTED; }
    const JsonValue* ty = pt->get("type");
    const std::string type = ty && ty->kind == JsonValue::Str ? ty->str : "";
    if (type == "Sequence") {
        const JsonValue* l = pt->get("pretokenizers");
        if (!l || l->kind != JsonValue::Arr) { cz_set_error("tokenizer.json: malformed Sequence pre_tokenizer"); return CZ_ERR_IO; }
        for (auto& e : l->arr) {
            if (*seen_bytelevel) { cz_set_error("tokenizer.json: a pre_tokenizer after ByteLevel is not supported"); return CZ_ERR_UNSUPPORTED; }
            CZCHK(parse_pretok(&e, t, seen_bytelevel));
        }
        return CZ_OK;
    }
    if (type == "Digits") {
        if (t->digits) { cz_set_error("tokenizer.json: two Digits pre_tokenizers"); return CZ_ERR_UNSUPPORTED; }
        const JsonValue* v = pt->get("individual_digits");
        t->digits = true; t->individual_digits = v && v->kind == JsonValue::Bool && v->b;
        return CZ_OK;
    }
    if (type == "ByteLevel") {
        const JsonValue* aps = pt->get("add_prefix_space"); const JsonValue* ur = pt->get("use_regex");
        if (aps && aps->kind == JsonValue::Bool && aps->b) { cz_set_error("tokenizer.json: ByteLevel add_prefix_space = true is not supported"); return CZ_ERR_UNSUPPORTED; }
        t->use_regex = !(ur && ur->kind == JsonValue::Bool && !ur->b);
        *seen_bytelevel = true;
        return CZ_OK;
    }
    cz_set_error("tokenizer.json: pre_tokenizer type '%s' is not supported", type.c_str());
    return CZ_ERR_UNSUPPORTED;
}

extern "C" cz_status cz_bpe_tokenizer_open(const char* path, cz_bpe_tokenizer** out) {
    if (!path || !out) { cz_set_error("cz_bpe_tokenizer_open: null argument"); return CZ_ERR_ARG; }
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { cz_set_error("cannot open %s", path); return CZ_ERR_IO; }
    std::string txt; char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, n);
    fclose(f);
    JsonValue root; std::string err;
    if (!cz_json_parse(txt.data(), txt.size(), &root, &err) || root.kind != JsonValue::Obj) { cz_set_error("tokenizer.json: %s", err.c_str()); return CZ_ERR_IO; }
    const JsonValue* norm = root.get("normalizer");
    if (norm && norm->kind != JsonValue::Null) { cz_set_error("tokenizer.json: a normalizer is not supported"); return CZ_ERR_UNSUPPORTED; }
    const JsonValue* model = root.get("model");
    const JsonValue* mty = model ? model->get("type") : nullptr;
    if (!model || model->kind != JsonValue::Obj || !mty || mty->kind != JsonValue::Str || mty->str != "BPE") { cz_set_error("tokenizer.json: model.type must be BPE"); return CZ_ERR_UNSUPPORTED; }
    for (const char* key : {"continuing_subword_prefix", "end_of_word_suffix", "dropout"}) {
        const JsonValue* v = model->get(key);
        if (v && v->kind != JsonValue::Null && !(v->kind == JsonValue::Str && v->str.empty())) { cz_set_error("tokenizer.json: model.%s is not supported", key); return CZ_ERR_UNSUPPORTED; }
    }
    const JsonValue* bf = model->get("byte_fallback");
    if (bf && bf->kind == JsonValue::Bool && bf->b) { cz_set_error("tokenizer.json: byte_fallback is not supported"); return CZ_ERR_UNSUPPORTED; }
    std::unique_ptr<cz_bpe_tokenizer> t(new cz_bpe_tokenizer());
    const JsonValue* im = model->get("ignore_merges");
    t->ignore_merges = im && im->kind == JsonValue::Bool && im->b;
    bool seen_bl = false;
    CZCHK(parse_pretok(root.get("pre_tokenizer"), t.get(), &seen_bl));
    if (!seen_bl) { cz_set_error("tokenizer.json: no ByteLevel pre_tokenizer"); return CZ_ERR_UNSUPPORTED; }
    const JsonValue* dec = root.get("decoder"); const JsonValue* dty = dec && dec->kind == JsonValue::Obj ? dec->get("type") : nullptr;
    if (!dty || dty->kind != JsonValue::Str || dty->str != "ByteLevel") { cz_set_error("tokenizer.json: decoder must be ByteLevel"); return CZ_ERR_UNSUPPORTED; }
    // vocab
    const JsonValue* vocab = model->get("vocab");
    if (!vocab || vocab->kind != JsonValue::Obj) { cz_set_error("tokenizer.json: model.vocab missing"); return CZ_ERR_IO; }
    std::unordered_map<std::string, uint32_t> text2id;            // token text as written in the file
    uint32_t max_id = 0;
    for (auto& kv : vocab->obj) { if (kv.second.kind != JsonValue::Num) continue; const uint32_t id = (uint32_t)kv.second.num; text2id[kv.first] = id; if (id > max_id) max_id = id; }
    const JsonValue* added = root.get("added_tokens");
    if (added && added->kind == JsonValue::Arr)
        for (auto& a : added->arr) { const JsonValue* id = a.get("id"); if (id && id->kind == JsonValue::Num && (uint32_t)id->num > max_id) max_id = (uint32_t)id->num; }
    t->id2bytes.assign((size_t)max_id + 1, std::string()); t->is_special.assign((size_t)max_id + 1, 0);
    for (auto& kv : text2id) {
        std::string raw;
        if (!token_to_bytes(kv.first, &raw)) raw = kv.first;      // not in the byte alphabet (an added token listed in the vocab): verbatim
        t->id2bytes[kv.second] = raw;
        t->bytes2id[raw] = kv.second;
    }
    for (int b = 0; b < 256; ++b) {
        std::string one(1, (char)b);
        auto it = t->bytes2id.find(one);
        t->byte_id[b] = it == t->bytes2id.end() ? -1 : (int32_t)it->second;
    }
    if (added && added->kind == JsonValue::Arr)
        for (auto& a : added->arr) {
            const JsonValue *id = a.get("id"), *c = a.get("content"), *sp = a.get("special");
            if (!id || id->kind != JsonValue::Num || !c || c->kind != JsonValue::Str || c->str.empty()) continue;
            for (const char* flag : {"single_word", "lstrip", "rstrip", "normalized"}) {
                const JsonValue* v = a.get(flag);
                if (v && v->kind == JsonValue::Bool && v->b) { cz_set_error("tokenizer.json: added token '%s' with %s = true is not supported", c->str.c_str(), flag); return CZ_ERR_UNSUPPORTED; }
            }
            const bool special = sp && sp->kind == JsonValue::Bool && sp->b;
            t->added.push_back({c->str, (uint32_t)id->num, special});
            t->id2bytes[(uint32_t)id->num] = c->str; t->is_special[(uint32_t)id->num] = special ? 1 : 0;
        }
    // merges: "a b" strings (older files) or [a, b] pairs
    const JsonValue* merges = model->get("merges");
    if (!merges || merges->kind != JsonValue::Arr) { cz_set_error("tokenizer.json: model.merges missing"); return CZ_ERR_IO; }
    uint32_t rank = 0;
    for (auto& m : merges->arr) {
        std::string a, b;
        if (m.kind == JsonValue::Str) { const size_t sp = m.str.find(' '); if (sp == std::string::npos) { cz_set_error("tokenizer.json: malformed merge"); return CZ_ERR_IO; } a = m.str.substr(0, sp); b = m.str.substr(sp + 1); }
        else if (m.kind == JsonValue::Arr && m.arr.size() == 2 && m.arr[0].kind == JsonValue::Str && m.arr[1].kind == JsonValue::Str) { a = m.arr[0].str; b = m.arr[1].str; }
        else { cz_set_error("tokenizer.json: malformed merge"); return CZ_ERR_IO; }
        auto ia = text2id.find(a), ib = text2id.find(b), iab = text2id.find(a + b);
        if (ia == text2id.end() || ib == text2id.end() || iab == text2id.end()) { cz_set_error("tokenizer.json: merge '%s %s' refers to tokens outside the vocabulary", a.c_str(), b.c_str()); return CZ_ERR_IO; }
        t->merges.emplace(std::make_pair(ia->second, ib->second), std::make_pair(rank, iab->second));
        ++rank;
    }
    *out = t.release();
    return CZ_OK;
}
extern "C" void cz_bpe_tokenizer_close(cz_bpe_tokenizer* t) { delete t; }
extern "C" size_t cz_bpe_vocab_size(const cz_bpe_tokenizer* t) { return t ? t->id2bytes.size() : 0; }    // get_vocab_size(true)

// merges of one pre-token: lowest rank first, leftmost first (the order HF's Word::merge_all pops its queue in)
static cz_status bpe_word(const cz_bpe_tokenizer* t, const std::string& piece, std::vector<uint32_t>* ids) {
    auto hit = t->cache.find(piece);
    if (hit != t->cache.end()) { ids->insert(ids->end(), hit->second.begin(), hit->second.end()); return CZ_OK; }
    std::vector<uint32_t> res;
    auto whole = t->bytes2id.find(piece);
    if (t->ignore_merges && whole != t->bytes2id.end()) res.push_back(whole->second);
    else {
        const size_t n = piece.size();
        std::vector<uint32_t> sym(n); std::vector<int> prev(n), next(n); std::vector<uint8_t> alive(n, 1);
        for (size_t i = 0; i < n; ++i) {
            const int32_t id = t->byte_id[(uint8_t)piece[i]];
            if (id < 0) { cz_set_error("byte 0x%02x has no token in the vocabulary", (unsigned)(uint8_t)piece[i]); return CZ_ERR_FORMAT; }
            sym[i] = (uint32_t)id; prev[i] = (int)i - 1; next[i] = i + 1 < n ? (int)i + 1 : -1;
        }
        struct Cand { uint32_t rank; int pos; uint32_t a, b, merged; };
        auto cmp = [](const Cand& x, const Cand& y) { return x.rank != y.rank ? x.rank > y.rank : x.pos > y.pos; };
        std::priority_queue<Cand, std::vector<Cand>, decltype(cmp)> pq(cmp);
        auto push = [&](int i) {
            if (i < 0 || next[i] < 0) return;
            auto it = t->merges.find(std::make_pair(sym[i], sym[next[i]]));
            if (it != t->merges.end()) pq.push({it->second.first, i, sym[i], sym[next[i]], it->second.second});
        };
        for (size_t i = 0; i + 1 < n; ++i) push((int)i);
        while (!pq.empty()) {
            const Cand c = pq.top(); pq.pop();
            const int i = c.pos;
            if (!alive[i] || next[i] < 0 || sym[i] != c.a || sym[next[i]] != c.b) continue;      // stale entry
            const int j = next[i];
            sym[i] = c.merged; alive[j] = 0; next[i] = next[j];
            if (next[j] >= 0) prev[next[j]] = i;
            push(prev[i]); push(i);
        }
        for (int i = n ? 0 : -1; i >= 0; i = next[i]) res.push_back(sym[i]);
    }
    if (t->cache.size() < (1u << 20)) t->cache.emplace(piece, res);
    ids->insert(ids->end(), res.begin(), res.end());
    return CZ_OK;
}

// GPT-2 pattern over code points [i, n): end of the first alternative that matches at i
static size_t gpt2_match(const uint32_t* c, size_t i, size_t n) {
    if (c[i] == '\'' && i + 1 < n) {
        const uint32_t x = c[i + 1], y = i + 2 < n ? c[i + 2] : 0;
        if (x == 's' || x == 't' || x == 'm' || x == 'd') return i + 2;
        if ((x == 'r' && y == 'e') || (x == 'v' && y == 'e') || (x == 'l' && y == 'l')) return i + 3;
    }
    const size_t j0 = (c[i] == ' ' && i + 1 < n) ? i + 1 : i;      // after the optional space (only useful if something follows)
    for (int alt = 0; alt < 3; ++alt) {
        auto in_class = [&](uint32_t ch) { return alt == 0 ? is_letter(ch) : alt == 1 ? is_number(ch) : !(is_white(ch) || is_letter(ch) || is_number(ch)); };
        size_t j = j0;
        if (in_class(c[j])) { while (j < n && in_class(c[j])) ++j; return j; }
        if (j0 != i && in_class(c[i])) { j = i; while (j < n && in_class(c[j])) ++j; return j; }   // ' ?' matched empty (c[i] == ' ' is never in a class: kept for symmetry)
    }
    // white space: \s+(?!\S) then \s+
    size_t j = i;
    while (j < n && is_white(c[j])) ++j;
    if (j == i) return i + 1;                                       // unreachable: every code point belongs to one of the classes above
    if (j == n || j - i == 1) return j;                             // run reaches the end, or a single space before a word (the \s+ alternative)
    return j - 1;                                                   // leave the last white-space character to prefix the next word
}

static cz_status encode_plain(const cz_bpe_tokenizer* t, const uint32_t* cps, size_t n, std::vector<uint32_t>* ids) {
    // Digits(individual_digits) first, then the byte-level pattern inside every remaining span
    size_t i = 0;
    std::string piece;
    auto emit_span = [&](size_t a, size_t b) -> cz_status {
        size_t p = a;
        while (p < b) {
            const size_t e = t->use_regex ? gpt2_match(cps + a, p - a, b - a) + a : b;
            piece.clear();
            for (size_t k = p; k < e; ++k) put_utf8(cps[k], &piece);
            CZCHK(bpe_word(t, piece, ids));
            p = e;
        }
        return CZ_OK;
    };
    while (i < n) {
        if (t->digits && is_number(cps[i])) {
            size_t j = i + 1;
            if (!t->individual_digits) while (j < n && is_number(cps[j])) ++j;
            CZCHK(emit_span(i, j));
            i = j;
        } else {
            size_t j = i;
            while (j < n && !(t->digits && is_number(cps[j]))) ++j;
            CZCHK(emit_span(i, j));
            i = j;
        }
    }
    return CZ_OK;
}

// tok.encode(text, add_special_tokens = false): added tokens are cut out of the raw text first (leftmost, longest)
extern "C" cz_status cz_bpe_encode(const cz_bpe_tokenizer* t, const uint8_t* utf8, size_t n, uint32_t** ids_out, size_t* n_ids) {
    if (!t || (!utf8 && n) || !ids_out || !n_ids) { cz_set_error("cz_bpe_encode: null argument"); return CZ_ERR_ARG; }
    std::vector<uint32_t> cps; cps.reserve(n);
    decode_utf8_lossy(utf8, n, &cps);
    std::vector<std::vector<uint32_t>> added_cps(t->added.size());
    for (size_t a = 0; a < t->added.size(); ++a) decode_utf8_lossy((const uint8_t*)t->added[a].content.data(), t->added[a].content.size(), &added_cps[a]);
    std::vector<uint32_t> ids;
    size_t start = 0, i = 0;
    while (i < cps.size()) {
        int best = -1; size_t best_len = 0;
        for (size_t a = 0; a < added_cps.size(); ++a) {
            const auto& w = added_cps[a];
            if (w.empty() || w[0] != cps[i] || w.size() > cps.size() - i || w.size() <= best_len) continue;
            if (std::equal(w.begin(), w.end(), cps.begin() + i)) { best = (int)a; best_len = w.size(); }
        }
        if (best >= 0) {
            CZCHK(encode_plain(t, cps.data() + start, i - start, &ids));
            ids.push_back(t->added[best].id);
            i += best_len; start = i;
        } else ++i;
    }
    CZCHK(encode_plain(t, cps.data() + start, cps.size() - start, &ids));
    uint32_t* o = (uint32_t*)malloc((ids.size() ? ids.size() : 1) * sizeof(uint32_t));
    if (!o) { cz_set_error("out of memory"); return CZ_ERR_ARG; }
    if (!ids.empty()) memcpy(o, ids.data(), ids.size() * sizeof(uint32_t));
    *ids_out = o; *n_ids = ids.size();
    return CZ_OK;
}

// tokenizer.decode(ids, skip_special_tokens): bytes of the tokens, then String::from_utf8_lossy like the ByteLevel decoder
extern "C" cz_status cz_bpe_decode(const cz_bpe_tokenizer* t, const uint32_t* ids, size_t n, int skip_special, uint8_t** out, size_t* out_len) {
    if (!t || (!ids && n) || !out || !out_len) { cz_set_error("cz_bpe_decode: null argument"); return CZ_ERR_ARG; }
    std::string raw;
    for (size_t i = 0; i < n; ++i) {
        if (ids[i] >= t->id2bytes.size()) { cz_set_error("token id %u outside the vocabulary (%zu)", ids[i], t->id2bytes.size()); return CZ_ERR_ARG; }
        if (skip_special && t->is_special[ids[i]]) continue;
        raw += t->id2bytes[ids[i]];
    }
    std::vector<uint32_t> cps; cps.reserve(raw.size());
    decode_utf8_lossy((const uint8_t*)raw.data(), raw.size(), &cps);
    std::string txt; txt.reserve(raw.size());
    for (uint32_t c : cps) put_utf8(c, &txt);
    uint8_t* o = (uint8_t*)malloc(txt.size() ? txt.size() : 1);
    if (!o) { cz_set_error("out of memory"); return CZ_ERR_ARG; }
    if (!txt.empty()) memcpy(o, txt.data(), txt.size());
    *out = o; *out_len = txt.size();
    return CZ_OK;
}

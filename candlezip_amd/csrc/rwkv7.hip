// rwkv7.hip — RWKV7 "Goose" recurrent-state path (placeholder until the kernels land; every entry fails loudly).
#include "cz_host.h"
#include "cz_session.h"
struct Rwkv7Model { int dummy; };
cz_status cz_rwkv7_build(cz_session*, const cz_model_desc*) { cz_set_error("rwkv7 backend: kernels not built in this revision"); return CZ_ERR_UNSUPPORTED; }
void cz_rwkv7_destroy(cz_session*) {}
size_t cz_rwkv7_vocab(const cz_session*) { return 0; }
cz_status cz_rwkv7_step_host(cz_session*, uint32_t, float*) { cz_set_error("rwkv7 backend unavailable"); return CZ_ERR_UNSUPPORTED; }
cz_status cz_rwkv7_reprime_host(cz_session*, const uint32_t*, size_t, float*) { cz_set_error("rwkv7 backend unavailable"); return CZ_ERR_UNSUPPORTED; }
cz_status cz_rwkv7_encode(cz_session*, size_t, const uint32_t* const*, const size_t*, uint32_t* const*, uint32_t* const*) { cz_set_error("rwkv7 backend unavailable"); return CZ_ERR_UNSUPPORTED; }
cz_status cz_rwkv7_decode_group(cz_session*, int, const uint8_t* const*, const size_t*, const size_t*, uint32_t, uint32_t* const*) { cz_set_error("rwkv7 backend unavailable"); return CZ_ERR_UNSUPPORTED; }

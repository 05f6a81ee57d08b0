#!/usr/bin/env python3
"""Tuning aid: segment-parallel encode only (no decode), SmolLM2-135M shape: `streams` streams x `tokens` tokens; prints tokens/s and the
per-family HIP-event split.  Used under rocprofv3 (--pmc / --kernel-trace) to look at the big-M GEMM and the prefill attention alone."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=1)
if len(sys.argv) > 4:
    s.debug_set("attn_v1", int(sys.argv[4]))      # 0 = attention_v2 (default), 1 = the LDS-resident v1 kernel
for kv in os.environ.get("CZ_BENCH_DEBUG_SET", "").split(","):      # e.g. CZ_BENCH_DEBUG_SET=gemm_pipe=0
    if "=" in kv:
        s.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
if os.environ.get("CZ_ATTN_DBG"):                   # tuning build only (make attn_dbg; CANDLEZIP_AMD_LIB=.../libcandlezip_hip_attndbg.so): results are wrong
    cz.lib().cz_debug_attn_dbg(int(os.environ["CZ_ATTN_DBG"]))
rng = np.random.default_rng(3)
streams = [np.concatenate([[0], rng.integers(0, 49152, T)]).astype(np.uint32) for _ in range(B)]
s.encode_tokens_batch(streams[:8])            # page the code in
s.profile_enable(True)
t0 = time.perf_counter()
for _ in range(reps):
    s.encode_tokens_batch(streams)
dt = time.perf_counter() - t0
prof = s.profile_read()
print(json.dumps({"streams": B, "tokens": T, "compress_tokens_per_s": round(reps * B * T / dt, 1),
                  "families_ms": {k: round(v["total_ms"], 1) for k, v in prof.items()},
                  "families_tflops": {k: round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 1) for k, v in prof.items() if v["flops"] > 0},
                  "families_launches": {k: v["launches"] for k, v in prof.items()},
                  "gemm_tflops": round(sum(v["flops"] for k, v in prof.items() if k.startswith("gemm")) / max(sum(v["total_ms"] for k, v in prof.items() if k.startswith("gemm")), 1e-9) / 1e9, 1),
                  "attention_tflops": round(prof.get("attention_f32_mfma", {"flops": 0})["flops"] / max(prof.get("attention_f32_mfma", {"total_ms": 1e-9})["total_ms"], 1e-9) / 1e9, 1)}))
if os.environ.get("CZ_ATTN_DBG") is not None and hasattr(cz.lib(), "cz_debug_attn_stamps"):
    import ctypes as C
    o = (C.c_ulonglong * 16)()
    cz.lib().cz_debug_attn_stamps(o, 0)
    nb, nc = max(o[0], 1), max(o[8], 1)
    print(json.dumps({"attention_v2_phase_cycles_per_block": {"sampled_blocks": int(o[0]), "chunks_per_block": round(nc / nb, 1), "q_and_first_k": round(o[1] / nb), "phase1_scores": round(o[2] / nb),
          "phase2_max": round(o[3] / nb), "phase3_exp_sums": round(o[4] / nb), "sums_combined": round(o[5] / nb), "phase4_p_pv": round(o[6] / nb), "stores": round(o[7] / nb),
          "phase1_per_chunk": round(o[2] / nc), "phase1_wait_k_registers": round(o[9] / nc), "phase1_stage_and_barrier": round(o[10] / nc), "phase1_reads_mfma_masks": round(o[11] / nc), "phase3_per_chunk": round(o[4] / nc), "phase4_per_chunk": round(o[6] / nc)}}))
s.close()

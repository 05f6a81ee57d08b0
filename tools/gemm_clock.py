#!/usr/bin/env python3
"""The clock the chip holds inside the big-M GEMM (tuning build `make -C candlezip_amd/csrc gemm_clock`):
    CANDLEZIP_AMD_LIB=candlezip_amd/lib/libcandlezip_hip_gemmclock.so python tools/gemm_clock.py [streams] [tokens]
Runs a segment-parallel encode for a few seconds (the kernels' steady state), then reads the shader-clock / 100 MHz tick sums that thread 0 of every 64th GEMM block
accumulated over its K loop: clock = shader ticks / real ticks x 100 MHz.  Prints the clock, the f32 MFMA peak AT that clock (1024 SIMDs x 64 flop/clk) and the
HIP-event TFLOP/s of the GEMM families beside it."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
L = cz.lib()
if not hasattr(L, "cz_debug_gemm_clock"):
    sys.exit("this library has no clock stamps: build `make gemm_clock` and set CANDLEZIP_AMD_LIB")
s = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=1)
rng = np.random.default_rng(3)
streams = [np.concatenate([[0], rng.integers(0, 49152, T)]).astype(np.uint32) for _ in range(B)]
if len(sys.argv) > 3:
    s.debug_set("gemm_lds_kb", int(sys.argv[3]))     # e.g. 100: one GEMM block per CU
s.encode_tokens_batch(streams)                       # warm: clocks settle under load
out = (C.c_ulonglong * 32)()
L.cz_debug_gemm_clock(out, 1)
s.profile_enable(True)
t0 = time.perf_counter()
for _ in range(2):
    s.encode_tokens_batch(streams)
dt = time.perf_counter() - t0
prof = s.profile_read()
L.cz_debug_gemm_clock(out, 0)
ghz = (out[0] + out[8] + out[16]) / max(out[1] + out[9] + out[17], 1) * 0.1
def fam(o, pipe):
    nb, nkt = max(out[o + 2], 1), max(out[o + 3], 1)
    blk = (out[o + 4] + out[o] + out[o + 5]) / nb
    return {"sampled_blocks": int(out[o + 2]), "k_tiles_per_block": round(nkt / nb, 2), "k_loop_cycles_per_k_tile": round(out[o] / nkt, 1), "mfma_pipe_cycles_per_k_tile": pipe,
            "cycles_entry_to_first_tile_staged": round(out[o + 4] / nb, 1), "cycles_k_loop_end_to_stores_issued": round(out[o + 5] / nb, 1), "cycles_per_block": round(blk, 1),
            "mfma_pipe_cycles_per_block": round(pipe * nkt / nb, 1)}
fl = sum(v["flops"] for k, v in prof.items() if k.startswith("gemm_bf16w")); ms = sum(v["total_ms"] for k, v in prof.items() if k.startswith("gemm_bf16w"))
print(json.dumps({"streams": B, "tokens": T, "clock_ghz_inside_gemm_k_loop": round(ghz, 3),
                  "f32_mfma_peak_at_that_clock_tflops": round(1024 * 64 * ghz / 1e3, 1), "nominal_peak_tflops": 157.3,
                  "gemm_tflops_hip_events": round(fl / ms / 1e9, 1), "frac_of_peak_at_that_clock": round(fl / ms / 1e9 / (1024 * 64 * ghz / 1e3), 3),
                  "tiles_128x128_lm_head": fam(0, 8192), "lm_head_k_tile_sections_cycles": {"mfma_section": round(out[24] / max(out[3], 1), 1), "wait_barrier_1": round(out[25] / max(out[3], 1), 1), "staging_stores": round(out[26] / max(out[3] - out[2], 1), 1), "wait_barrier_2": round(out[27] / max(out[3] - out[2], 1), 1)}, "tiles_128x128_gate_up_swiglu": fam(16, 8192), "tiles_128x64_qkv_o_down": fam(8, 4096), "gemm_lds_kb": int(sys.argv[3]) if len(sys.argv) > 3 else 0,
                  "compress_tokens_per_s": round(2 * B * T / dt, 1)}))
s.close()

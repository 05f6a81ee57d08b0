#!/usr/bin/env python3
"""Where gemm_mid_kernel spends its time in the wide decode step (tuning build `make -C candlezip_amd/csrc mid_dbg`):
    CANDLEZIP_AMD_LIB=candlezip_amd/lib/libcandlezip_hip_middbg.so python tools/mid_phases.py [B] [pos] [iters]
Wave 0 of the first and of the last block of every launch stamps s_memrealtime at its phase boundaries (each stamp waits for the wave's outstanding
memory operations first); mean per launch in microseconds, by epilogue kind (plain = o / down, swiglu = gate|up, rope = QKV)."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 767
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = cz.lib()
if not hasattr(lib, "cz_debug_mid_stamps"):
    sys.exit("this library has no phase stamps: build `make mid_dbg` and set CANDLEZIP_AMD_LIB")
lib.cz_debug_decode_steps.restype = C.c_int
lib.cz_debug_decode_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
s = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=B)
s.debug_set("wide_min_streams", 256); s.debug_set("wide_mode", 3)
out = C.c_double()
assert lib.cz_debug_decode_steps(s.h, B, pos, 0, 1, 1, C.byref(out)) == 0, lib.cz_last_error()
buf = (C.c_ulonglong * 128)()
lib.cz_debug_mid_stamps(buf, 1)
assert lib.cz_debug_decode_steps(s.h, B, pos, iters, 1, 1, C.byref(out)) == 0, lib.cz_last_error()
lib.cz_debug_mid_stamps(buf, 0)
names = {0: "plain (o / down), first block", 1: "swiglu (gate|up), first block", 2: "rope (QKV), first block", 4: "plain, last block", 5: "swiglu, last block", 6: "rope, last block"}
ph = {1: "first tile arrived", 2: "partials of previous round", 3: "staging + MFMA chain", 4: "barrier", 5: "last partials", 15: "epilogue"}
res = {"B": B, "pos": pos, "ms_per_step": round(out.value / iters, 4), "kernels": {}}
for k, nm in names.items():
    n = buf[k * 16]
    if n:
        res["kernels"][nm] = {"launches_per_step": round(n / iters, 1), **{ph[p]: round(buf[k * 16 + p] * 0.01 / n, 2) for p in ph},
                              "in_kernel_total_us": round(sum(buf[k * 16 + p] for p in ph) * 0.01 / n, 2)}
print(json.dumps(res, indent=1))
s.close()

#!/usr/bin/env python3
"""Where the narrow-group decode kernels spend their time, phase by phase (tuning build `make -C candlezip_amd/csrc dec_dbg`):
    CANDLEZIP_AMD_LIB=candlezip_amd/lib/libcandlezip_hip_decdbg.so python tools/dec_phases.py [B] [pos] [iters]
Block (0, 0) / thread 0 of every dec_gemm_kernel / dec_attn_kernel launch stamps s_memrealtime at its phase boundaries (each stamp waits for the
wave's outstanding memory operations first); the table is the mean per launch in microseconds.  Shares, not absolute truths: the waits forbid
overlaps the real kernel has."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 767
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = cz.lib()
if not hasattr(lib, "cz_debug_dec_stamps"):
    sys.exit("this library has no phase stamps: build `make dec_dbg` and set CANDLEZIP_AMD_LIB")
lib.cz_debug_decode_steps.restype = C.c_int
lib.cz_debug_decode_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
s = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=max(B, 1))
out = C.c_double()
assert lib.cz_debug_decode_steps(s.h, B, pos, 0, 1, 1, C.byref(out)) == 0, lib.cz_last_error()
buf = (C.c_ulonglong * 128)()
lib.cz_debug_dec_stamps(buf, 1)
assert lib.cz_debug_decode_steps(s.h, B, pos, iters, 1, 1, C.byref(out)) == 0, lib.cz_last_error()
lib.cz_debug_dec_stamps(buf, 0)
names = {0: "plain (o / down / lm_head)", 1: "qkv+rope (no norm)", 2: "gate|up (no norm)", 3: "dec_attn", 4: "plain+norm", 5: "qkv+rope+norm", 6: "gate|up+norm", 7: "relusq+norm"}
ph_gemm = {1: "operands arrived", 2: "norm phase", 3: "MFMA chains", 4: "ordered reduction", 15: "epilogue"}
ph_attn = {1: "desc/pos/Q", 2: "K stream + scores", 3: "max barrier", 4: "exp + chunk sums", 5: "V stream + P.V", 15: "final sum + out"}
res = {"B": B, "pos": pos, "ms_per_step": round(out.value / iters, 4), "kernels": {}}
for k in range(8):
    n = buf[k * 16]
    if not n:
        continue
    ph = ph_attn if k == 3 else ph_gemm
    res["kernels"][names[k]] = {"launches_per_step": round(n / iters, 1), **{ph[p]: round(buf[k * 16 + p] * 0.01 / n, 2) for p in ph},
                                "in_kernel_total_us": round(sum(buf[k * 16 + p] for p in ph) * 0.01 / n, 2)}
print(json.dumps(res, indent=1))
s.close()

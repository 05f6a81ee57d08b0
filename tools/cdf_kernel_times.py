#!/usr/bin/env python3
"""Tuning aid: launches of the exact-parallel CDF kernels alone, for `rocprofv3 --kernel-trace` (tools/prof_summary.py then gives their durations):
encode form on `rows` rows for the SmolLM (49 152, mode 0) and the RWKV (65 536 + 256 literals, mode 1) alphabets, and the
one-row decode form (a single stream's step).  Usage: cdf_kernel_times.py [rows]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = cz.lib()
rng = np.random.default_rng(0)
for V, mode in ((49152, 0), (65536, 1)):
    logits = (rng.standard_normal((rows, V)) * 4).astype(np.float32)
    n = V + (256 if mode else 0)
    sym = rng.integers(0, n, rows).astype(np.uint32)
    lo = np.zeros(rows, np.uint32); hi = np.zeros(rows, np.uint32); fb = np.zeros(1, np.uint32)
    for rep in range(3):
        rc = L.cz_debug_cdf_pairs(C.c_void_p(logits.ctypes.data), C.c_void_p(sym.ctypes.data), C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data), rows, V, mode, 0, C.c_void_p(fb.ctypes.data))
        assert rc == 0 and fb[0] == 0
    # one stream's decode step: 8 symbols of one row sequence
    steps = 8
    lg = (rng.standard_normal((steps, V)) * 4).astype(np.float32)
    pay = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
    out = np.zeros(steps, np.uint32)
    rc = L.cz_debug_decode_symbols(C.c_void_p(lg.ctypes.data), steps, V, mode, pay, len(pay), C.c_void_p(out.ctypes.data), 0)
    assert rc == 0
    print(f"V={V} mode={mode}: ok", flush=True)

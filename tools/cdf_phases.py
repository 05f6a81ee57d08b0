#!/usr/bin/env python3
"""Per-phase time inside the exact-parallel CDF kernel (tuning aid).  Needs the timers build:
    make -C candlezip_amd/csrc timers && CANDLEZIP_AMD_LIB=candlezip_amd/lib/libcandlezip_hip_timers.so python tools/cdf_phases.py [V] [mode] [rows]
Phases (block 0, thread 0, wall_clock64 at 100 MHz): 0 load+exp, 1..4 the sequential chains, 5 final walk; inside a chain:
8 approx scan, 9 transducer chains, 10 copies, 11 segmented scan, 12 ordered resolve."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

V = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 128
decode = len(sys.argv) > 4 and sys.argv[4] == "decode"     # the DECODE instantiation: ONE stream, `rows` steps (cz_debug_decode_symbols) on random payload bytes
L = cz.lib()
rng = np.random.default_rng(0)
logits = (rng.standard_normal((rows, V)) * 4).astype(np.float32)
n = V + (256 if mode else 0)
sym = rng.integers(0, n, rows).astype(np.uint32)
lo = np.zeros(rows, np.uint32); hi = np.zeros(rows, np.uint32); fb = np.zeros(1, np.uint32)
out = (C.c_ulonglong * 16)()
if decode:
    payload = rng.integers(0, 256, rows * 4 + 64).astype(np.uint8)
    dout = np.zeros(rows, np.uint32)
    L.cz_debug_decode_symbols.restype = C.c_int
    for rep in range(3):
        L.cz_debug_cdf_phases(out, 1)
        rc = L.cz_debug_decode_symbols(C.c_void_p(logits.ctypes.data), rows, V, mode, C.c_void_p(payload.ctypes.data), C.c_size_t(payload.size), C.c_void_p(dout.ctypes.data), 0)
        assert rc == 0, L.cz_last_error()
        L.cz_debug_cdf_phases(out, 0)
    t = [x / 100.0 / rows for x in out]           # us per step (the timers accumulate over the steps)
    print(f"DECODE V={V} mode={mode} steps={rows}: per step {sum(t[:6]):.1f} us = load+exp {t[0]:.1f} | chains {t[1]:.1f} {t[2]:.1f} {t[3]:.1f} {t[4]:.1f} | symbol search + decoder update {t[5]:.1f}")
    print(f"  inside the chains: scan {t[8]:.1f}  transducers {t[9]:.1f}  copies {t[10]:.1f}  seg-scan {t[11]:.1f}  resolve {t[12]:.1f}")
    sys.exit(0)
for rep in range(3):
    L.cz_debug_cdf_phases(out, 1)
    rc = L.cz_debug_cdf_pairs(C.c_void_p(logits.ctypes.data), C.c_void_p(sym.ctypes.data), C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data),
                              rows, V, mode, 0, C.c_void_p(fb.ctypes.data))
    assert rc == 0
    L.cz_debug_cdf_phases(out, 0)
t = [x / 100.0 for x in out]           # us
print(f"V={V} mode={mode} rows={rows}: row total {sum(t[:6]):.1f} us = load+exp {t[0]:.1f} | chains {t[1]:.1f} {t[2]:.1f} {t[3]:.1f} {t[4]:.1f} | final {t[5]:.1f}")
print(f"  inside the chains (summed): scan {t[8]:.1f}  transducers {t[9]:.1f}  copies {t[10]:.1f}  seg-scan {t[11]:.1f}  resolve {t[12]:.1f}  (fallbacks {fb[0]})")

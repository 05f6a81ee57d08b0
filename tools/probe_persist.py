#!/usr/bin/env python3
"""What one all-to-all hand-off costs INSIDE a launch on this device (cz_debug_probe_persist): G resident blocks, every stage = request the stage's
weights, wait for all blocks of the previous stage (agent-scope counter), read the 576-float activation vector they wrote, write the own slice,
fence, count in.  stride 8 = only the blocks the dispatcher deals to ONE XCD take part (one L2).  One JSON line per shape; compare with
tools/probe_chain.py (the same dependency as kernel boundaries of a replayed graph)."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

lib = cz.lib()
lib.cz_debug_probe_persist.restype = C.c_int
lib.cz_debug_probe_persist.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
stages = 2000
for G, stride, kb, xn in ((256, 1, 0, 576), (256, 1, 4, 576), (256, 1, 16, 576), (128, 1, 0, 576), (128, 1, 8, 576), (128, 1, 36, 576), (64, 1, 0, 576), (64, 1, 16, 576),
                          (32, 1, 0, 576), (32, 1, 36, 576), (16, 1, 0, 576), (8, 1, 0, 576), (2, 1, 0, 576),
                          (32, 8, 0, 576), (32, 8, 8, 576), (32, 8, 36, 576), (64, 8, 0, 576), (64, 8, 16, 576), (16, 8, 0, 576), (8, 8, 0, 576),
                          (256, 1, 4, 1536), (32, 8, 36, 1536)):
    us = C.c_double()
    rc = lib.cz_debug_probe_persist(G, stride, stages, kb, xn, 3, C.byref(us))
    print(json.dumps({"blocks": G, "one_xcd": stride == 8, "weights_kb_per_block_and_stage": kb, "weights_mb_per_stage": round(G * kb / 1024, 2), "vector_floats": xn,
                      "us_per_stage": round(us.value, 3) if rc == 0 else None, "error": None if rc == 0 else lib.cz_last_error().decode()}), flush=True)

#!/usr/bin/env python3
"""The floor under a chain of dependent kernels on this device (cz_debug_probe_chain): microseconds per node of a 155-node chain — the shape of a
single stream's decode step — for nodes that do nothing, and for nodes that stream what a layer's GEMVs stream.  One JSON line per shape."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

lib = cz.lib()
lib.cz_debug_probe_chain.restype = C.c_int
lib.cz_debug_probe_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for name, blocks, kb in (("nothing, 1 block", 1, 0), ("nothing, 256 blocks", 256, 0), ("18 blocks x 36 KB (o-projection: 0.66 MB)", 18, 36),
                         ("96 blocks x 36 KB (gate|up: 3.5 MB)", 96, 36), ("256 blocks x 16 KB (4 MB)", 256, 16), ("1024 blocks x 64 KB (64 MB: lm_head-sized)", 1024, 64)):
    out = {}
    for graph in (1, 0):
        us = C.c_double()
        rc = lib.cz_debug_probe_chain(155, blocks, kb, 40, graph, C.byref(us))
        out["graph" if graph else "eager"] = round(us.value, 2) if rc == 0 else None
    print(json.dumps({"node": name, "us_per_node": out}), flush=True)

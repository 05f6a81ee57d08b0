#!/usr/bin/env python3
"""RWKV7-0.1B-shaped self-test throughput (BASELINE config 4: recurrent-state kernel path), B streams in lock-step.
Usage: bench_rwkv7.py [streams] [tokens]   -> one JSON line"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = cz.rwkv7_0b1_config()
LANE = int(sys.argv[3]) if len(sys.argv) > 3 else B
s = cz.Session("rwkv7", cfg, seed=0x5EED, embed_std=0.02, max_streams=min(B, LANE))
s.debug_set("max_lanes", int(sys.argv[4]) if len(sys.argv) > 4 else 4)
for kv in filter(None, os.environ.get("CZ_BENCH_DEBUG_SET", "").split(",")):       # e.g. CZ_BENCH_DEBUG_SET=gemv_max_rows=0
    k, v = kv.split("="); s.debug_set(k, int(v))
rng = np.random.default_rng(1)
streams = [np.concatenate([[0], rng.integers(0, cfg.vocab, T)]).astype(np.uint32) for _ in range(B)]
for rep in range(2):                       # first pass warms up
    t0 = time.perf_counter()
    pairs = s.encode_tokens_batch(streams)
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    t1 = time.perf_counter()
    outs = s.decode_tokens_batch(pays, [T] * B)
    t2 = time.perf_counter()
assert all(np.array_equal(a, b) for a, b in zip(streams, outs))
n = B * T
print(json.dumps({"backend": "rwkv7", "shape": "RWKV7-0.1B (V65536 C768 L12)", "streams": B, "tokens_per_stream": T,
                  "compress_tokens_per_s": round(n / (t1 - t0), 1), "decompress_tokens_per_s": round(n / (t2 - t1), 1),
                  "self_test_tokens_per_s": round(n / (t2 - t0), 1), "debug_set": os.environ.get("CZ_BENCH_DEBUG_SET", ""), "bits_per_token_synthetic": round(8 * sum(len(p) for p in pays) / n, 3)}))

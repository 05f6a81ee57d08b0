#!/usr/bin/env python3
"""Host-side cost of replaying the decode step graph (tuning aid): wall and thread-CPU time inside hipGraphLaunch, and the cost of a
launch into an empty queue (first 64 steps are synchronised before each launch)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = cz.smollm2_135m_config()
s = cz.Session("smollm", cfg, seed=0x5EED, embed_std=0.02, max_streams=B)
for kv in sys.argv[3:]:                      # extra debug knobs: key=value
    k, v = kv.split("="); s.debug_set(k, int(v))
rng = np.random.default_rng(0)
streams = [np.concatenate([[0], rng.integers(0, cfg.vocab, T)]).astype(np.uint32) for _ in range(B)]
pairs = s.encode_tokens_batch(streams)
pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
for probe in (0, 1, 0):
    s.debug_set("probe_launch", probe)
    t0 = time.perf_counter()
    outs = s.decode_tokens_batch(pays, [T] * B)
    dt = time.perf_counter() - t0
    print(f"probe={probe} B={B} T={T}: decode {dt:.3f}s ({dt / T * 1e3:.3f} ms/step); in hipGraphLaunch wall {s.debug_get('graph_launch_us') / T:.1f} us/step, "
          f"cpu {s.debug_get('graph_launch_cpu_us') / T:.1f} us/step, idle-queue launches {s.debug_get('graph_launch_idle_us') / 64:.1f} us each", flush=True)
assert all(np.array_equal(a, b) for a, b in zip(streams, outs))

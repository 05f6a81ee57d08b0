#!/usr/bin/env python3
"""The pass bench.py measures its rooflines on, alone, for rocprofv3 (put THIS program directly after `--`):

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_pass.py smollm
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/profile_pass.py smollm      (again with WRITE_SIZE)
    ... python3 tools/profile_pass.py rwkv7

smollm: one lock-step group of `--streams` (512) streams x `--tokens` (2048) tokens, context 512 / re-prime 512 (or --context / --reprime-interval):
the encode of all streams, then the lock-step decode — eager launches, one host synchronisation per decode step (library knob step_sync: the
profiler serialises dispatches, a deep queue makes it crawl).  tools/prof_summary.py --skip-frac 0.25 then drops every kernel's first quarter of
dispatches: for the decode-step kernels that is the ramp-up 1 -> 512 keys, what is left is the steady-state sawtooth 511 -> 1023 bench.py's timed
region lives in.  rwkv7: `--streams` (1024: the bench's group width) streams x `--tokens` (256) tokens: sequence-parallel encode, then the lock-step decode.
No torch is imported: the process loads ONE ROCm runtime (the one libcandlezip_hip.so links).  A heartbeat line goes to stderr every 30 s."""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("backend", choices=["smollm", "rwkv7"])
    ap.add_argument("--streams", type=int, default=None)
    ap.add_argument("--tokens", type=int, default=None)
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--reprime-interval", type=int, default=512)
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE")
    a = ap.parse_args()
    t0 = time.perf_counter()
    stop = threading.Event()

    def beat():
        while not stop.wait(30.0):
            print(f"[profile_pass +{time.perf_counter() - t0:6.0f}s] running", file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()
    if a.backend == "smollm":
        S, T = a.streams or 512, a.tokens or 2048
        cfg = cz.smollm2_135m_config()
        s = cz.Session("smollm", cfg, seed=0x5EED, embed_std=0.02, max_streams=S)
    else:
        S, T = a.streams or 1024, a.tokens or 256
        cfg = cz.rwkv7_0b1_config()
        s = cz.Session("rwkv7", cfg, seed=0x5EED, embed_std=0.02, max_streams=S)
    s.debug_set("max_lanes", 1); s.debug_set("no_graph", 1); s.debug_set("step_sync", 1)
    if a.backend == "smollm":
        s.debug_set("dec_attn_waves", 4)           # the block shape bench.py's two lanes run (one group alone would pick eight-wave blocks); --debug-set dec_attn_waves=8 for those
    for kv in a.debug_set:
        k, v = kv.split("=")
        s.debug_set(k, int(v))
    rng = np.random.default_rng(99)
    streams = [np.concatenate([[0], rng.integers(0, cfg.vocab, T)]).astype(np.uint32) for _ in range(S)]
    if a.backend == "smollm":
        pairs = s.encode_tokens_batch(streams, a.context, a.reprime_interval)
        pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
        outs = s.decode_tokens_batch(pays, [T] * S, a.context, a.reprime_interval, 0)
    else:
        pairs = s.encode_tokens_batch(streams)
        pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
        outs = s.decode_tokens_batch(pays, [T] * S)
    ok = all(np.array_equal(x, y) for x, y in zip(streams, outs))
    stop.set()
    print(f"[profile_pass] {a.backend}: {S} streams x {T} tokens, round trip {'ok' if ok else 'MISMATCH'}, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
    s.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — candlezip hot path on MI355X: tokens/sec of the enwik8-style self-test (compress + decompress).

Workload = BASELINE.json configs[2] ("enwik8 64 KiB samples --backend smollm, chunk-sharded"), SmolLM2-135M shape, context 512 /
reprime 512: per GPU `--streams` persistent streams of `--sample-tokens` tokens (a 64 KiB enwik8 sample is ~17.4k SmolLM2 tokens).
No checkpoint / tokenizer / enwik8 exists offline, so weights are seeded synthetic bf16 tensors and ids are uniform random:
throughput and roofline are meaningful, bits/byte is not (reported as null).

One "step" = one `--window`-token slice of the self-test of EVERY stream, as a two-stage pipeline on the same persistent streams: the
segment-parallel compress of the next window of the compress stage (cz_encode_tokens_range_batch + the serial arithmetic coder fed window
by window) and the lock-step decompress of the next window of the decompress stage, which runs two windows behind and is fed the coder's
bytes as they become final (cz_decode_feed_batch / cz_decode_advance: KV cache, arithmetic-decoder state and decoded ids stay in HBM
between steps).  Streams are whole 17 408-token samples, so the re-prime sawtooth every step sees is the steady-state one whatever K is:
17 408 / window steps are exactly one 64 KiB sample per stream, and any other K times the same configuration (the window is a multiple of
the re-prime interval: a smaller one would re-compute the 511 history rows of a segment once per window).  Every sample's round trip is
verified when its last window has been decoded.  Pipeline fill (two windows of compress) happens before the warmup steps and is not timed.

Contract line (rank 0): metric/value/unit/n_gpus/steps/warmup/ms_per_step/... + "roofline" (the kernel FUNCTION with the largest
total time, all its instantiations summed; HIP-event timed on the library's stream in an eager pass inside this script) +
"roofline_dec_attention", "roofline_mfma_gemm" + "cpu_baseline" (oracle, reference schedule, bounded sample, rank 0 at N=1 only) +
"single_stream" (rank 0 at N=1: ONE stream alone on the device, as the reference CLI runs a file — an untimed extra, never part of `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak BW (spec)


def note(msg):
    """progress on stderr (long runs must not look hung to the GPU harness)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()

# kernel FUNCTION -> (profile families of cz_profile_read, bound, rocprofv3 kernel-name prefix)
FUNCTIONS = {
    "dec_gemm_kernel": (["dec_gemm_plain", "dec_gemm_qkv_rope", "dec_gemm_swiglu", "dec_gemm_lm_head"], "mfma", "dec_gemm_kernel<"),
    "dec_attn_kernel": (["dec_attention"], "hbm", "dec_attn_kernel<"),
    "dec_rmsnorm_kernel": (["dec_rmsnorm"], "hbm", "dec_rmsnorm_kernel"),
    "cdf_par_kernel (decode)": (["decode_symbol"], "hbm", "cdf_par_kernel<1"),
    "cdf_par_kernel (encode)": (["cdf_pairs"], "hbm", "cdf_par_kernel<0"),
    "gemm_bf16w_kernel": (["gemm_bf16w_f32_mfma", "gemm_bf16w_f32_mfma_n64", "gemm_skinny_f32_mfma"], "mfma", "gemm_bf16w_kernel<"),
    "attention_v2_kernel": (["attention_f32_mfma"], "mfma", "attention_v2_kernel<"),
    "rmsnorm_kernel": (["rmsnorm"], "hbm", "rmsnorm_kernel"), "rope_kernel": (["rope"], "hbm", "rope_kernel"),
    "swiglu_kernel": (["swiglu"], "hbm", "swiglu_kernel"), "embed_kernel": (["embed"], "hbm", "embed_kernel"),
}


def pmc_traffic(prefix, tokens=2048):
    """HBM bytes per launch of a kernel function from the committed rocprofv3 --pmc passes (profiles/r*_pmc*_{FETCH,WRITE}_SIZE_pmc.json,
    made by tools/prof_summary.py from separate FETCH_SIZE / WRITE_SIZE runs).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    counts 128-B requests of wide coalesced reads at 64 B, so it is doubled (MI355X_MICROARCH.md, "HBM").  The newest file that holds
    the kernel wins.  Returns (bytes_per_launch or None, note)."""
    import glob
    tot = 0.0; notes = []
    for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
        # the pass of the same length as the eager profiling pass first (r<round>_pmc<tokens>_...), newest round first; then any other
        # (a pass of the OTHER backend never: both hold gemm_mid_kernel / gemm_bf16w_kernel, on different shapes)
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc{tokens}_{counter}_pmc.json")), reverse=True)
        if str(tokens).isdigit():
            files += sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc[0-9]*_{counter}_pmc.json")), reverse=True)
        hit = None
        for f in files:
            d = json.load(open(f))
            ks = [k for k in d if k.startswith(prefix) and counter in d[k]]
            if ks:
                hit = (f, sum(d[k][counter]["dispatches"] for k in ks), sum(d[k][counter]["sum"] for k in ks))
                break
        if not hit or not hit[1]:
            notes.append(f"{counter}: no pass committed for {prefix}")
            continue
        tot += hit[2] * scale / hit[1]
        notes.append(f"{counter} from {os.path.basename(hit[0])} ({hit[1]} dispatches{', x2 gfx950 correction' if counter == 'FETCH_SIZE' else ''})")
    return (round(tot) if tot else None), "; ".join(notes)


def rocprof_avg_us(prefix, tokens):
    """average kernel duration of a kernel function from the committed rocprofv3 --kernel-trace --stats summary of the pass this script
    measures its rooflines on (profiles/r*_pass<tokens>_*kernel_stats.csv, tools/prof_summary.py; newest round first).  The HIP-event figure
    next to it brackets launch + kernel + the gap to the next event record, so it reads a few us longer per launch."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pass{tokens}_*kernel_stats.csv")), reverse=True)
    for f in files:
        calls = 0; tot = 0.0
        for line in open(f).read().splitlines()[1:]:             # kernel names hold commas (template arguments): the six numbers are the LAST six fields
            parts = line.rsplit(",", 6)
            if len(parts) == 7 and parts[0].startswith(prefix):
                calls += int(parts[1]); tot += float(parts[2])
        if calls:
            return round(1e3 * tot / calls, 3), os.path.basename(f)
    return None, None


class Generation:
    """one 64 KiB sample per stream: ids, one incremental arithmetic encoder per stream, the payload bytes not yet handed to the decoder"""

    def __init__(self, cz, stream_ids, gen, n_tokens, vocab):
        self.ids = []
        for sid in stream_ids:                                   # seeded by the GLOBAL stream id: the work list is the same for any rank count
            rng = np.random.default_rng([1000 + sid, gen])
            self.ids.append(np.concatenate([[0], rng.integers(0, vocab, n_tokens)]).astype(np.uint32))
        self.encs = [cz.AcEncoder() for _ in stream_ids]
        self.pending = [b"" for _ in stream_ids]
        self.complete = False
        self.bytes = 0

    def take(self):
        out, self.pending = self.pending, [b"" for _ in self.pending]
        return out


class SelfTestPipeline:
    """The self-test of persistent streams as a two-stage pipeline over 512-token windows: the compress stage runs LAG windows ahead of the
    decompress stage ON THE SAME STREAMS; the arithmetic coder's bytes are fed to the device decoder as they become final
    (cz_ac_encoder_push -> cz_decode_feed_batch), exactly as a decompressor reading a file that is still being written would see them."""
    LAG = 2

    def __init__(self, cz, sess, stream_ids, n_tokens, window, vocab, context, reprime, enc_sess=None):
        self.cz, self.sess, self.sids, self.T, self.W, self.V = cz, sess, stream_ids, n_tokens, window, vocab
        self.enc_sess = enc_sess      # a fork of the session (own HIP stream, same weights): the compress half then runs WHILE the decompress half does
        self.ctx, self.R = context, reprime
        self.nwin = -(-n_tokens // window)
        self.gens = {}
        self.cg = self.cw = self.dg = self.dw = 0                # compress / decompress cursors: (generation, window)
        self.te = self.td = 0.0; self.tok = 0; self.bytes = 0; self.verified = 0; self.windows_verified = 0

    def _window(self, w):
        return w * self.W, min((w + 1) * self.W, self.T)

    def _gen(self, g):
        if g not in self.gens:
            self.gens[g] = Generation(self.cz, self.sids, g, self.T, self.V)
        return self.gens[g]

    def _compress_next(self):
        g, w = self._gen(self.cg), self.cw
        b, e = self._window(w)
        t = time.perf_counter()
        parts = (self.enc_sess or self.sess).encode_tokens_range_batch(g.ids, [(b, e)] * len(g.ids), self.ctx, self.R)
        last = w == self.nwin - 1
        for k, (lo, hi) in enumerate(parts):                     # ArithmeticEncoder::encode_counts, window by window (src/main.rs:2300); finish (:2358)
            nb = g.encs[k].push(lo, hi)
            if last:
                nb += g.encs[k].finish()[0]
                g.encs[k].close()
            g.pending[k] += nb
            g.bytes += len(nb)
        g.complete = g.complete or last
        self.cw += 1
        if self.cw == self.nwin:
            self.cg += 1; self.cw = 0
        return time.perf_counter() - t

    def _begin_decode(self):
        g = self._gen(self.dg)
        self.sess.decode_begin_partial_batch(g.take(), [4 * self.T + 64] * len(self.sids), [self.T] * len(self.sids), self.ctx, self.R, 0)
        self._fed_final = False

    def _feed(self):
        g = self._gen(self.dg)
        if not self._fed_final and (any(g.pending) or g.complete):
            self.sess.decode_feed_batch(g.take(), final=g.complete)
            self._fed_final = g.complete

    def fill(self):
        """untimed: the compress stage gets LAG windows ahead, the decode of generation 0 is opened"""
        for _ in range(self.LAG):
            self._compress_next()
        self._begin_decode()

    def _check(self, upto):
        outs = self.sess.decode_fetch_batch()                    # (raises if a decoder ran past the bytes fed so far)
        for s, o in zip(self._gen(self.dg).ids, outs):
            if not np.array_equal(s[: upto + 1], o[: upto + 1]):
                raise RuntimeError("self-test round trip mismatch")

    def step(self):
        """compress the next window of the compress stage + decompress the next window of the decompress stage"""
        b, e = self._window(self.dw)
        if self.enc_sess is not None:                            # two jobs share the device: MFMA-bound compress beside the HBM-bound decompress
            import threading
            self._feed()
            box = {}

            def work():
                try:
                    box["te"] = self._compress_next()
                except BaseException as ex:                      # re-raised on the main thread below: never a KeyError that hides it
                    box["err"] = ex
            th = threading.Thread(target=work)
            th.start()
            t = time.perf_counter()
            try:
                self.sess.decode_advance(e - b)
            finally:
                self.td += time.perf_counter() - t
                th.join()
            if "err" in box:
                raise box["err"]
            self.te += box["te"]
        else:
            self.te += self._compress_next()
            t = time.perf_counter()
            self._feed()
            self.sess.decode_advance(e - b)
            self.td += time.perf_counter() - t
        self.tok += (e - b) * len(self.sids)
        self.dw += 1
        if self.dw == self.nwin:                                 # generation complete: verify, move on to the next one
            t = time.perf_counter()
            self._check(self.T)
            self.verified += 1; self.windows_verified += self.nwin
            self.bytes += self._gen(self.dg).bytes
            self.sess.decode_end()
            del self.gens[self.dg]
            self.dg += 1; self.dw = 0
            self._begin_decode()
            self.td += time.perf_counter() - t

    def finish(self):
        """verify what has been decoded of the generation in flight, release the slot"""
        self._check(self._window(self.dw)[0] if self.dw else 0)
        self.windows_verified += self.dw
        self.sess.decode_end()


def profile_pass(cz, sess, streams, context, reprime, skip):
    """the eager per-family pass on STEADY-STATE contexts: the first `skip` tokens of every stream — the ramp-up 1 -> 512 keys, which a
    17 408-token stream sees during 1/34 of its life — run with the timers off; the timed part is whole re-prime periods of the sawtooth the
    timed region lives in (511 -> 1023 keys at the defaults, mean 767).  Encode: the range [skip, n) (whole segments when skip is a multiple of
    the re-prime interval); decode: the sliced loop, timers switched on after the first slice.  Round trip verified.  Returns the merged
    family table."""
    n = [s.size - 1 for s in streams]
    sess.profile_enable(False)
    pairs = sess.encode_tokens_batch(streams, context, reprime)
    payloads = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    sess.profile_enable(True)
    part = sess.encode_tokens_range_batch(streams, [(min(skip, k), k) for k in n], context, reprime)
    prof = sess.profile_read()
    sess.profile_enable(False)
    for (lo, hi), (plo, phi), k in zip(pairs, part, n):
        if not (np.array_equal(lo[min(skip, k):], plo) and np.array_equal(hi[min(skip, k):], phi)):
            raise RuntimeError("range encode differs from the one-shot encode")
    sess.debug_set("no_graph", 1)                                # eager launches: every family can be bracketed by events
    sess.decode_begin_batch(payloads, n, context, reprime, 0)
    if skip:
        sess.decode_advance(skip)
    sess.profile_enable(True)
    sess.decode_advance(max(n))
    dec = sess.profile_read()
    sess.profile_enable(False)
    outs = sess.decode_fetch_batch()
    sess.decode_end()
    sess.debug_set("no_graph", 0)
    for s, o in zip(streams, outs):
        if not np.array_equal(s, o):
            raise RuntimeError("self-test round trip mismatch")
    for k, v in dec.items():
        if k in prof:
            prof[k] = {f: prof[k][f] + v[f] for f in v}
        else:
            prof[k] = v
    return prof


def single_stream_line(cz, cfg, tokens, context, reprime, backend="smollm"):
    """BASELINE configs[1] as the reference CLI runs it: ONE stream (one file) alone on the device, compress then decompress, round trip checked.  A short
    stream on a session of its own, untimed part of the run: the figure beside the many-stream `value` (narrow groups on VALU projections)."""
    s1 = cz.Session(backend, cfg, seed=0x5EED, embed_std=0.02, max_streams=1)
    try:
        rng = np.random.default_rng(4)
        ids = np.concatenate([[0], rng.integers(0, cfg.vocab, tokens)]).astype(np.uint32)
        s1.encode_tokens(ids[:600], context, reprime)                    # page the code in
        t0 = time.perf_counter()
        lo, hi = s1.encode_tokens(ids, context, reprime)
        pay, zw = cz.ac_encode_pairs(lo, hi)
        t1 = time.perf_counter()
        out = s1.decode_tokens(pay, tokens, context, reprime)
        t2 = time.perf_counter()
        ok = bool(zw or np.array_equal(out, ids))
    finally:
        s1.close()
    return {"workload": f"ONE stream x {tokens} tokens (one file = one stream, as the reference CLI runs it)" + (f", context {context}, reprime_interval {reprime}" if backend == "smollm" else ", recurrent state (no re-prime)"),
            "compress_tokens_per_s": round(tokens / (t1 - t0), 1), "decompress_tokens_per_s": round(tokens / (t2 - t1), 1),
            "self_test_tokens_per_s": round(tokens / (t2 - t0), 1), "round_trip_exact": ok, "zero_width_steps": int(zw)}


def cpu_baseline(sample_tokens, context, reprime):
    """the oracle (CPU restatement of the reference path) in the REFERENCE's schedule: one M=1 forward per token, a
    511-token batched re-prime every `reprime` tokens, host softmax->CDF->coder per token.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import czo
    note("cpu baseline: oracle in the reference schedule ...")
    o = czo.SmolLmSession(czo.smollm2_135m_config(), seed=0x5EED, embed_std=0.02)
    rng = np.random.default_rng(7)
    ids = np.concatenate([[0], rng.integers(0, 49152, sample_tokens)]).astype(np.uint32)
    t0 = time.perf_counter()
    payload, clo, chi, zw = o.encode_tokens(ids, context, reprime)
    t1 = time.perf_counter()
    out = o.decode_tokens(payload, sample_tokens, context, reprime, 0)
    t2 = time.perf_counter()
    assert np.array_equal(out, ids)
    cores = czo.lib().czo_num_threads()
    o.close()
    return {"value": round(sample_tokens / (t2 - t0), 2), "unit": "tokens/s", "cores": int(cores), "kind": "port",
            "sample": f"self-test (compress {t1 - t0:.1f}s + decompress {t2 - t1:.1f}s) of one {sample_tokens}-token stream, SmolLM2-135M shape, "
                      f"ctx {context}/reprime {reprime}, f32, OpenMP oracle",
            "compress_tokens_per_s": round(sample_tokens / (t1 - t0), 2), "decompress_tokens_per_s": round(sample_tokens / (t2 - t1), 2)}


def spawn_ranks(n, argv=None, script=None):
    """`python bench.py --gpus N` without a launcher around it: start N fresh child ranks, one per GPU, BEFORE anything in this process has
    touched the GPU (children are new processes; nothing that initialised HIP is ever exec'd).  Rank 0's stdout — the ONE JSON line — is
    this process's stdout; the exit code is the worst child's.  A rank that dies takes the others down (they would wait at a barrier)."""
    import socket
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC only on this pool (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or (code if code > 0 else 128 - code)
                for q in live:                                   # exact PIDs of our own children
                    q.terminate()
        time.sleep(0.2)
    return rc


def cpu_baseline_rwkv7(sample_tokens):
    """the oracle's RWKV7-0.1B walk (one T == 1 forward per token, host CDF with literals, coder), bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import czo
    note("cpu baseline: RWKV7 oracle ...")
    o = czo.Rwkv7Session(czo.rwkv7_0b1_config(), seed=0x5EED, embed_std=0.02)
    rng = np.random.default_rng(7)
    ids = np.concatenate([[0], rng.integers(0, 65536, sample_tokens)]).astype(np.uint32)
    t0 = time.perf_counter()
    payload, clo, chi, zw = o.encode_tokens(ids)
    t1 = time.perf_counter()
    out = o.decode_tokens(payload, sample_tokens)
    t2 = time.perf_counter()
    assert np.array_equal(out, ids)
    cores = czo.lib().czo_num_threads()
    o.close()
    return {"value": round(sample_tokens / (t2 - t0), 2), "unit": "tokens/s", "cores": int(cores), "kind": "port",
            "sample": f"self-test (compress {t1 - t0:.1f}s + decompress {t2 - t1:.1f}s) of one {sample_tokens}-token stream, RWKV7-0.1B shape, f32, OpenMP oracle",
            "compress_tokens_per_s": round(sample_tokens / (t1 - t0), 2), "decompress_tokens_per_s": round(sample_tokens / (t2 - t1), 2)}


def main_rwkv7(args):
    """BASELINE configs[3]: `--backend rwkv7` on Canterbury-sized inputs, 1 GPU, recurrent-state kernel path.  One step = the self-test of a fresh
    batch: `--streams` streams (chunks of the corpus) x `--sample-tokens` tokens each — compress (sequence-parallel: every token of a time
    block through the projections at once, only the WKV scan sequential in t) + the serial coder + decompress (lock-step groups on decode lanes) —
    round trip verified every step.  RWKV has no re-prime: a stream's state starts at zero, like a file's.  Weights / ids synthetic (no
    checkpoint, vocabulary or corpus offline): bits/byte is null."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CZ_BENCH_DIST_BACKEND", "nccl")
        if backend == "gloo":
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import candlezip_amd as cz
    cfg = cz.tiny_rwkv7_config() if args.tiny else cz.rwkv7_0b1_config()
    S, T = args.streams, args.sample_tokens
    groups = max(1, min(args.lanes, -(-S // args.lane_streams)))
    per_lane = min(1024, -(-S // groups))
    sess = cz.Session("rwkv7", cfg, seed=0x5EED, embed_std=0.02, device=local, max_streams=per_lane)
    sess.debug_set("max_lanes", groups)
    if args.no_graph:
        sess.debug_set("no_graph", 1)
    for kv in args.debug_set:
        k, v = kv.split("=")
        sess.debug_set(k, int(v))
    note(f"rank {rank}/{world}: rwkv7 session open, {S} streams x {T} tokens, {groups} lanes x {per_lane}")

    def batch(step):
        rng = np.random.default_rng([4000 + rank, step + 1000])
        return [np.concatenate([[0], rng.integers(0, cfg.vocab, T)]).astype(np.uint32) for _ in range(S)]

    te = td = 0.0; nbytes = 0
    # --overlap 1 (default): a two-stage pipeline as on the smollm line — the compress of batch k + 1 runs on a forked session (own HIP stream and work buffers,
    # shared weights) WHILE batch k is decompressed: the MFMA-bound sequence-parallel encode beside the lock-step decode.  Every batch is both compressed and
    # decompressed; K steps are K compress halves and K decompress halves (the first batch's compress is the untimed pipeline fill, the last step's is of a batch
    # whose decompress would be step K + 1).  --overlap 0: one after the other, each half timed alone.
    enc_sess = sess.fork() if args.overlap else None
    import threading

    def compress(se, step):
        streams = batch(step)
        t0 = time.perf_counter()
        pairs = se.encode_tokens_batch(streams)
        pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
        return streams, pays, time.perf_counter() - t0

    def decompress(streams, pays):
        t0 = time.perf_counter()
        outs = sess.decode_tokens_batch(pays, [T] * S)
        dt = time.perf_counter() - t0
        for a, b in zip(streams, outs):
            if not np.array_equal(a, b):
                raise RuntimeError("self-test round trip mismatch")
        return dt

    pending = [None]

    def one(step):
        nonlocal te, td, nbytes
        if enc_sess is None:
            streams, pays, dte = compress(sess, step)
            dtd = decompress(streams, pays)
        else:
            if pending[0] is None:
                pending[0] = compress(enc_sess, step)                        # pipeline fill (warmup / untimed)
            streams, pays, _ = pending[0]
            box = {}
            th = threading.Thread(target=lambda: box.update(r=compress(enc_sess, step + 1)))
            th.start()
            dtd = decompress(streams, pays)
            th.join()
            if "r" not in box:
                raise RuntimeError("the compress half failed")
            pending[0] = box["r"]; dte = box["r"][2]
        te += dte; td += dtd; nbytes += sum(len(p) for p in pays)

    def barrier():
        sess.synchronize()
        if dist is not None:
            dist.barrier()
            sess.synchronize()

    sess.profile_enable(False)
    for w in range(max(args.warmup, 1 if enc_sess is not None else 0)):      # (the pipeline needs one untimed step to fill)
        one(-100 + w)
    te = td = 0.0; nbytes = 0
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one(k)
        note(f"step {k + 1}/{args.steps}: compress {te:.1f}s decompress {td:.1f}s")
    barrier()
    elapsed = time.perf_counter() - t0
    tok = args.steps * S * T
    # eager per-family pass: one lock-step group alone, every family bracketed by HIP events
    prof = {}
    if args.profile_tokens > 0:
        pt = min(args.profile_tokens, T)
        rng = np.random.default_rng(99)
        short = [np.concatenate([[0], rng.integers(0, cfg.vocab, pt)]).astype(np.uint32) for _ in range(min(per_lane, S, 1024))]       # (the timed region's own group width; 1 024 streams x 256 tokens: the pass profiles/r03_passrwkv7_s1024_t256_* were taken on)
        sess.debug_set("no_graph", 1)
        sess.profile_enable(True)
        pairs = sess.encode_tokens_batch(short)
        pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
        outs = sess.decode_tokens_batch(pays, [pt] * len(short))
        prof = sess.profile_read()
        sess.profile_enable(False)
        sess.debug_set("no_graph", 1 if args.no_graph else 0)
        if not all(np.array_equal(a, b) for a, b in zip(short, outs)):
            raise RuntimeError("self-test round trip mismatch (profile pass)")
        note("eager per-family pass done")
    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        c = torch.tensor([tok], dtype=torch.float64, device=dev); dist.all_reduce(c, op=dist.ReduceOp.SUM); total_tok = float(c.item())
    else:
        total_tok = float(tok)
    if rank == 0:
        def roof(fam, bound, prefix, kernel):
            st = prof.get(fam)
            if not st or not st["total_ms"]:
                return None
            ms, n = st["total_ms"], max(st["launches"], 1)
            if bound == "mfma":
                ach = st["flops"] / (ms * 1e-3) / 1e12
                r = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4)}
            else:
                ach = st["bytes"] / (ms * 1e-3) / 1e9
                r = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
            traffic, tnote = pmc_traffic(prefix, "rwkv7")
            rp_us, rp_src = rocprof_avg_us(prefix, "rwkv7")
            if prefix.startswith("gemm_bf16w_kernel"):            # (launches of several shapes and epilogues: no per-launch comparison with the kernel-trace CSV)
                rp_us, rp_src = None, "not comparable per launch (launches of several shapes; the kernel-trace CSV holds the totals)"
            r.update({"traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": tnote, "kernel": kernel, "family": fam,
                      "algorithmic_per_launch": round((st["flops"] if bound == "mfma" else st["bytes"]) / n), "algorithmic_unit": "flop" if bound == "mfma" else "bytes",
                      "launches": st["launches"], "avg_launch_us": round(1e3 * ms / n, 3), "avg_kernel_us_rocprof": rp_us, "avg_kernel_us_rocprof_source": rp_src,
                      "total_ms": round(ms, 2), "measured_in": f"eager per-family pass inside bench.py ({min(per_lane, S, 1024)} streams x {min(args.profile_tokens, T)} tokens, one lane)"})
            if rp_us and st["launches"]:
                alg = (st["flops"] if bound == "mfma" else st["bytes"]) / n
                r["achieved_rocprof"] = round(alg / (rp_us * 1e-6) / (1e12 if bound == "mfma" else 1e9), 3)
                r["frac_rocprof"] = round(r["achieved_rocprof"] / (FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS), 4)
            if not st["launches"]:                                # --profile-tokens 0: nothing was measured; say so instead of printing zeros
                for k in ("achieved", "frac", "algorithmic_per_launch", "avg_launch_us", "total_ms"):
                    r[k] = None
                r["measured_in"] = "not measured: the per-family pass was skipped (--profile-tokens 0)"
            return r
        line = {
            "metric": "tokens/sec, Canterbury-style self-test (compress+decompress), rwkv7",
            "value": round(total_tok / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"--backend rwkv7, Canterbury-sized inputs as 64 KiB chunks: {S} streams x {T} tokens per GPU and step, one step = compress "
                                   f"(sequence-parallel) + coder + decompress (lock-step) of a fresh batch{' — the compress of batch k + 1 on a forked session beside the decompress of batch k' if args.overlap else ''}, RWKV7-0.1B shape ({'tiny test model' if args.tiny else 'C768 L12 H12x64 ffn3072 V65536 (+256 literals)'}), seeded synthetic weights",
                       "streams_per_gpu": S, "decode_lanes": groups, "streams_per_lane": per_lane, "tokens_per_stream": T, "compress_beside_decompress": bool(args.overlap), "parallelism": f"files x{world} (no data-path collective)"},
            "bits_per_byte": None, "bits_per_token_synthetic": round(8.0 * nbytes / max(tok, 1), 3),
            "steps_verified": args.steps, "ranks_seen": (dist.get_world_size() if dist is not None else 1),
            "compress_tokens_per_s_rank0": round(tok / te, 1) if te else None, "decompress_tokens_per_s_rank0": round(tok / td, 1) if td else None,
            "roofline": roof("rwkv7_state", "hbm", "rw_state_kernel", "rw_state_kernel (decode: WKV state update of one token per stream)"),
            "roofline_scan": roof("rwkv7_scan", "hbm", "rw_seq_scan_kernel", "rw_seq_scan_kernel (encode: WKV scan over a time block, state in registers)"),
            "roofline_rwkv_gemm_decode": roof("rwkv7_gemm", "mfma", "gemm_mid_", "gemm_mid_kernel / gemm_mid_group_kernel (projections of the wide decode step)"),
            "roofline_mfma_gemm": roof("gemm_bf16w_f32_mfma", "mfma", "gemm_bf16w_kernel<128, 128", "gemm_bf16w_kernel (encode projections, lm_head)"),
            "families_ms_profile_pass": {k: round(v["total_ms"], 2) for k, v in prof.items()},
        }
        if world == 1 and args.single_stream_tokens > 0 and not args.tiny:
            line["single_stream"] = single_stream_line(cz, cfg, min(args.single_stream_tokens, 2048), 512, 512, "rwkv7")
        if world == 1 and args.cpu_sample > 0 and not args.tiny:
            line["cpu_baseline"] = cpu_baseline_rwkv7(min(args.cpu_sample, 400))
        print(json.dumps(line), flush=True)
    if enc_sess is not None:
        enc_sess.close()
    sess.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="smollm", choices=["smollm", "rwkv7"], help="smollm (default: BASELINE configs[2], the driver's line) or rwkv7 (configs[3])")
    ap.add_argument("--steps", type=int, default=None, help="timed steps; 34 windows of 512 tokens are one 64 KiB sample per stream")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=None, help="streams (64 KiB chunks) per GPU, decoded in lock-step (KV cache: 94 MB per stream)")
    ap.add_argument("--lanes", type=int, default=None, help="concurrent lock-step decode groups per GPU (library decode lanes)")
    ap.add_argument("--lane-streams", type=int, default=None, help="streams per lock-step group (>= 256: the wide-group decode step)")
    ap.add_argument("--sample-tokens", type=int, default=None, help="tokens per stream (a 64 KiB enwik8 sample is ~17.4k SmolLM2 tokens)")
    ap.add_argument("--window", type=int, default=512, help="tokens per stream and step; a multiple of the re-prime interval, so that no window re-computes another one's history rows")
    ap.add_argument("--overlap", type=int, default=1, help="1 (default): the compress half of a step runs on a forked session WHILE the decompress half runs (+3-4 %: MFMA-bound beside HBM-bound); 0: one after the other, so that the two halves are timed alone")
    ap.add_argument("--no-graph", action="store_true", help="launch the decode step eagerly (one AQL packet per submission: for rocprofv3 --pmc)")
    ap.add_argument("--no-events", action="store_true", help="no HIP-event timing inside the timed region (for rocprofv3 --pmc)")
    ap.add_argument("--step-sync", action="store_true", help="with --no-graph: synchronise after every decode step, so that launches never queue up (rocprofv3 --pmc, DESIGN.md §9)")
    ap.add_argument("--kv-ctx-cap", type=int, default=0, help="tuning: positions per stream in the KV cache (0 = library default 2048)")
    ap.add_argument("--profile-tokens", type=int, default=None, help="tokens/stream of the untimed eager per-family profiling pass (0 = skip); its first --profile-skip tokens run with the timers off")
    ap.add_argument("--profile-skip", type=int, default=-1, help="tokens/stream at the start of the profiling pass that are not timed (the ramp-up to the steady-state sawtooth); -1 = one re-prime interval")
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--reprime-interval", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=640, help="tokens of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--single-stream-tokens", type=int, default=2560, help="tokens of the untimed single-stream measurement reported as `single_stream` (smollm line, one rank; 0 = skip)")
    ap.add_argument("--dec-priority", type=int, default=0, help="HIP stream priority of the decode lanes: -1 = high, 0 = normal")
    ap.add_argument("--enc-priority", type=int, default=0, help="HIP stream priority of the forked compress session (--overlap 1): 1 = low, 0 = normal, -1 = high")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="library tuning knob (cz_debug_set)")
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check only)")
    args = ap.parse_args()
    # defaults per backend: smollm = 1024 persistent streams x 17 408 tokens in 2 lock-step groups of 512, 17 window steps;
    # rwkv7 = 1024 streams x 2048 tokens per step in ONE lock-step group (the wide step's GEMMs fill the device best at 1024 rows: 179 k tokens/s
    # decompress against 168 k for 2 x 256, profiles/r03_rwkv7_lane_shapes_final.jsonl), 3 steps
    dflt = {"smollm": dict(steps=17, streams=1024, lanes=2, lane_streams=512, sample_tokens=17408, profile_tokens=2048),
            "rwkv7": dict(steps=3, streams=1024, lanes=1, lane_streams=1024, sample_tokens=2048, profile_tokens=256)}[args.backend]
    for k, v in dflt.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # no launcher: be one (before any GPU call in this process)
        sys.exit(spawn_ranks(args.gpus))

    if args.backend == "rwkv7":
        return main_rwkv7(args)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # One rank needs no torch at all: the product is libcandlezip_hip.so on the system ROCm runtime it links (ldd: /opt/rocm/lib/libamdhip64.so.7).
    # Importing torch first would put torch's BUNDLED libamdhip64 under this library instead — and under rocprofv3 that is a ROCm-7.0 HIP runtime
    # over the profiler's ROCm-7.2 HSA runtime in one process (the --pmc crash).  torch is plumbing for the N > 1 rendezvous only.
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CZ_BENCH_DIST_BACKEND", "nccl")       # "gloo": rehearsal of the N > 1 path on a one-GPU box
        if backend == "gloo":
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import candlezip_amd as cz
    from candlezip_amd.parallel import lpt_assign
    cfg = cz.tiny_llama_config() if args.tiny else cz.smollm2_135m_config()
    groups = max(1, min(args.lanes, -(-args.streams // args.lane_streams)))
    per_lane = min(1024, -(-args.streams // groups))                      # every lock-step group stays resident on a lane of its own
    sess = cz.Session("smollm", cfg, seed=0x5EED, embed_std=0.02, device=local, max_streams=per_lane)
    sess.debug_set("max_lanes", groups)
    if args.no_graph:
        sess.debug_set("no_graph", 1)
    if args.step_sync:
        sess.debug_set("step_sync", 1)
    if args.kv_ctx_cap:
        sess.debug_set("kv_ctx_cap", args.kv_ctx_cap)
    for kv in args.debug_set:
        k, v = kv.split("=")
        sess.debug_set(k, int(v))
    # work list: world x streams independent files; LPT over the ranks (equal sizes -> `streams` per rank), no data-path collective
    mine = lpt_assign([args.sample_tokens] * (world * args.streams), world)[rank]
    if args.dec_priority:
        sess.debug_set("stream_priority", args.dec_priority)
    enc_sess = sess.fork() if args.overlap else None
    if enc_sess is not None and args.enc_priority:
        enc_sess.debug_set("stream_priority", args.enc_priority)     # the throughput-bound encoder yields to the latency-bound decode lanes
    pipe = SelfTestPipeline(cz, sess, mine, args.sample_tokens, args.window, cfg.vocab, args.context, args.reprime_interval, enc_sess)
    note(f"rank {rank}/{world}: session open, {len(mine)} streams x {args.sample_tokens} tokens, {groups} lanes x {per_lane}, window {args.window}")
    if os.environ.get("CZ_DUMP_MAPS"):      # diagnostics: the address map of this process (to symbolise a native stack trace afterwards)
        import shutil
        shutil.copy("/proc/self/maps", os.environ["CZ_DUMP_MAPS"])

    def barrier():
        sess.synchronize()                                       # every stream of this rank's session: its lanes' and its forks' (the encoder)
        if dist is not None:
            dist.barrier()
            sess.synchronize()

    sess.profile_enable(False)
    pipe.fill()                                                           # untimed: the compress stage two windows ahead, the decode opened
    note("pipeline filled (compress stage two windows ahead)")
    if os.environ.get("CZ_DUMP_MAPS"):      # ... and again with every lane's HIP stream (its AQL ring) in existence
        import shutil
        shutil.copy("/proc/self/maps", os.environ["CZ_DUMP_MAPS"] + ".filled")
    for _ in range(args.warmup):                                          # W untimed steps of the same pipeline (first one builds the decode hipGraphs)
        pipe.step()
    pipe.te = pipe.td = 0.0; pipe.tok = 0

    # timed region: exactly K steps.  The big-M MFMA GEMM family (launched eagerly) is HIP-event timed on the library's
    # stream inside this region; the decode-step kernels run inside hipGraphs and cannot be bracketed one by one.
    if not args.no_events:
        sess.profile_enable(True, only=["gemm_bf16w_f32_mfma", "gemm_bf16w_f32_mfma_n64"])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.step()
        if k % 4 == 3 or k == args.steps - 1:
            note(f"step {k + 1}/{args.steps}: {pipe.tok} tokens so far, compress {pipe.te:.1f}s decompress {pipe.td:.1f}s, {pipe.verified} generation(s) verified")
    barrier()
    elapsed = time.perf_counter() - t0
    prof_timed = sess.profile_read() if not args.no_events else {}
    tok, te, td = pipe.tok, pipe.te, pipe.td
    pipe.finish()                                                          # round trip of the generation in flight, up to the decoded position

    # per-family pass (untimed, eager, every family HIP-event timed), one lock-step group alone on the device so that per-launch
    # durations are not inflated by the other lanes: names the dominant kernel FUNCTION and measures its achieved roofline
    prof_all = {}
    da_waves_knob = int(sess.debug_get("dec_attn_waves"))
    da_waves_timed = da_waves_knob if da_waves_knob else (4 if args.lanes > 1 else 8)     # (library: four-wave decode-attention blocks where lock-step groups share the device)
    if args.profile_tokens > 0:
        sess.debug_set("dec_attn_waves", da_waves_timed)                    # the pass runs ONE group alone: keep the block shape the timed region's lanes used
        sess.profile_enable(True)
        rng = np.random.default_rng(99)
        short = [np.concatenate([[0], rng.integers(0, cfg.vocab, args.profile_tokens)]).astype(np.uint32) for _ in range(min(per_lane, len(mine)))]
        skip = args.profile_skip if args.profile_skip >= 0 else min(args.reprime_interval, args.profile_tokens // 2)
        prof_all = profile_pass(cz, sess, short, args.context, args.reprime_interval, skip)
        sess.debug_set("dec_attn_waves", da_waves_knob)
        if args.no_graph:
            sess.debug_set("no_graph", 1)
        note("eager per-family pass done")
    sess.profile_enable(False)
    if dist is not None:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([tok], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_tok = float(c.item())
    else:
        total_tok = float(tok)

    if rank == 0:
        zero = {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0}

        def merged(prof, fams):
            return {k: sum(prof.get(f, zero)[k] for f in fams) for k in zero}

        def roof_of(fn, st, where):
            fams, bound, prefix = FUNCTIONS[fn]
            ms = st["total_ms"]
            if bound == "mfma":
                ach = st["flops"] / (ms * 1e-3) / 1e12 if ms else 0.0
                r = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4)}
            else:
                ach = st["bytes"] / (ms * 1e-3) / 1e9 if ms else 0.0
                r = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
            traffic, tnote = pmc_traffic(prefix, args.profile_tokens)
            n = max(st["launches"], 1)
            rp_us, rp_src = rocprof_avg_us(prefix, args.profile_tokens)
            if fn.startswith("gemm_bf16w_kernel"):                # its launches are not one shape: the batches of this pass (encode as a RANGE, whole re-primes) are cut differently from
                rp_us, rp_src = None, "not comparable per launch: tools/profile_pass.py cuts the same rows into other batches (the kernel-trace CSV holds its totals)"   # the profiled pass's
            r.update({"traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": tnote, "kernel": fn + " (all instantiations)", "families": fams,
                      "algorithmic_per_launch": round((st["flops"] if bound == "mfma" else st["bytes"]) / n), "algorithmic_unit": "flop" if bound == "mfma" else "bytes",
                      "launches": st["launches"], "avg_launch_us": round(1e3 * ms / n, 3),
                      "avg_launch_us_note": "HIP events around each launch on the library's stream: kernel + launch gap; `achieved` is computed from this figure",
                      "avg_kernel_us_rocprof": rp_us, "avg_kernel_us_rocprof_source": rp_src, "total_ms": round(ms, 2), "measured_in": where})
            if rp_us and st["launches"]:
                alg = (st["flops"] if bound == "mfma" else st["bytes"]) / n
                r["achieved_rocprof"] = round(alg / (rp_us * 1e-6) / (1e12 if bound == "mfma" else 1e9), 3)
                r["frac_rocprof"] = round(r["achieved_rocprof"] / (FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS), 4)
            if not st["launches"]:                                # --profile-tokens 0: nothing was measured; say so instead of printing zeros
                for k in ("achieved", "frac", "algorithmic_per_launch", "avg_launch_us", "total_ms"):
                    r[k] = None
                r["measured_in"] = "not measured: the per-family pass was skipped (--profile-tokens 0)"
            return r

        if per_lane >= 256:      # wide lock-step groups: the layer projections run on gemm_mid_kernel, lm_head and the norms on the big-M kernels
            FUNCTIONS.pop("dec_gemm_kernel"); FUNCTIONS.pop("dec_rmsnorm_kernel")
            FUNCTIONS["gemm_mid_kernel"] = (["dec_gemm_plain", "dec_gemm_qkv_rope", "dec_gemm_swiglu"], "mfma", "gemm_mid_kernel<")
            FUNCTIONS["gemm_bf16w_kernel (decode lm_head)"] = (["dec_gemm_lm_head"], "mfma", "gemm_bf16w_kernel<128, 128, 2, 2, 0>")
            FUNCTIONS["rmsnorm_kernel"] = (["rmsnorm", "dec_rmsnorm"], "hbm", "rmsnorm_kernel")
        where = (f"eager per-family pass inside bench.py ({min(per_lane, len(mine))} streams x {args.profile_tokens} tokens, the first "
                 f"{args.profile_skip if args.profile_skip >= 0 else min(args.reprime_interval, args.profile_tokens // 2)} untimed: whole periods of the steady-state context "
                 "sawtooth the timed region lives in; one decode lane); the timed region replays hipGraphs on concurrent lanes")
        by_fn = {fn: merged(prof_all, FUNCTIONS[fn][0]) for fn in FUNCTIONS}
        dominant = max(by_fn, key=lambda f: by_fn[f]["total_ms"]) if prof_all else "dec_attn_kernel"
        roof = roof_of(dominant, by_fn[dominant], where)
        roof["functions_ms_profile_pass"] = {f: round(v["total_ms"], 2) for f, v in by_fn.items() if v["total_ms"]}
        if dominant == "dec_attn_kernel":
            roof["block_shape"] = (f"{da_waves_timed} waves per block, as the timed region's {args.lanes} lane(s) ran it (round 4: four-wave blocks where lanes share the device: two lanes "
                                   "side by side 9.45 ms per step against 9.9 with eight-wave blocks, which a group ALONE runs 1-3 % faster: 0.69 of the HBM peak — DESIGN.md §9)")
        line = {
            "metric": "tokens/sec, enwik8 self-test (compress+decompress), smollm",
            "value": round(total_tok / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"enwik8 64 KiB samples, chunk-sharded: {args.streams} persistent streams x {args.sample_tokens} tokens per GPU, one step = a "
                                   f"{args.window}-token window of every stream (compress stage two windows ahead of the decompress stage, coder bytes fed as they "
                                   f"become final; {pipe.nwin} steps = one whole sample), SmolLM2-135M shape ({'tiny test model' if args.tiny else 'd576 L30 9q/3kv ffn1536 V49152'}), "
                                   f"context {args.context}, reprime_interval {args.reprime_interval}, seeded synthetic weights",
                       "streams_per_gpu": args.streams, "decode_lanes": groups, "streams_per_lane": per_lane, "tokens_per_stream": args.sample_tokens,
                       "window_tokens": args.window, "compress_beside_decompress": bool(args.overlap), "parallelism": f"files x{world} (LPT work list, no data-path collective)"},
            "bits_per_byte": None,
            "bits_per_token_synthetic": round(8.0 * pipe.bytes / max(pipe.verified * len(mine) * args.sample_tokens, 1), 3) if pipe.verified else None,
            "generations_verified": pipe.verified, "windows_verified": pipe.windows_verified,
            "ranks_seen": (dist.get_world_size() if dist is not None else 1), "dist_backend": (dist.get_backend() if dist is not None else None),
            "compress_tokens_per_s_rank0": round(tok / te, 1) if te else None,
            "decompress_tokens_per_s_rank0": round(tok / td, 1) if td else None,
            "halves_note": ("the two halves of a step ran at the same time and shared the device: each rate is lower than alone (--overlap 0 times them alone)"
                            if args.overlap else "the two halves of a step ran one after the other"),
            "roofline": roof,
        }
        if prof_all and dominant != "dec_attn_kernel":
            line["roofline_dec_attention"] = roof_of("dec_attn_kernel", by_fn["dec_attn_kernel"], where)
        dg = "gemm_mid_kernel" if "gemm_mid_kernel" in FUNCTIONS else "dec_gemm_kernel"
        if prof_all and dominant != dg:
            line["roofline_dec_gemm"] = roof_of(dg, by_fn[dg], where)
        if prof_all:                                             # the big-M GEMM of encode and re-prime: alone in the eager pass, and as it ran in the timed region
            line["roofline_mfma_gemm"] = roof_of("gemm_bf16w_kernel", by_fn["gemm_bf16w_kernel"], where)
            if prof_timed:
                t = merged(prof_timed, FUNCTIONS["gemm_bf16w_kernel"][0])
                if t["total_ms"]:
                    line["roofline_mfma_gemm"]["timed_region"] = {
                        "achieved": round(t["flops"] / (t["total_ms"] * 1e-3) / 1e12, 3), "unit": "TFLOP/s", "launches": t["launches"],
                        "note": "HIP events on decode lane 0 over the timed region: its re-prime launches, with the other lane and the encoder sharing the device"}
        if world == 1 and args.single_stream_tokens > 0 and not args.tiny:
            line["single_stream"] = single_stream_line(cz, cfg, args.single_stream_tokens, args.context, args.reprime_interval)
            note(f"single stream: {line['single_stream']['decompress_tokens_per_s']} tokens/s decompress")
        if world == 1 and args.cpu_sample > 0 and not args.tiny:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.context, args.reprime_interval)
        print(json.dumps(line), flush=True)
    if enc_sess is not None:
        enc_sess.close()
    sess.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

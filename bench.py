#!/usr/bin/env python3
"""bench.py — candlezip hot path on MI355X: tokens/sec of the enwik8-style self-test (compress + decompress).

One "step" = one self-test pass (segment-parallel encode -> arithmetic coding -> lock-step decode -> exact round-trip
check) of this rank's batch of synthetic streams, with token ids and weights resident in HBM-side buffers owned by the
library.  Workload = BASELINE.json configs[2] ("enwik8 64 KiB samples --backend smollm, chunk-sharded"), SmolLM2-135M
shape, context 512 / reprime 512: per GPU `--streams` streams of `--tokens` tokens (a 64 KiB enwik8 sample is ~17.4k
SmolLM2 tokens).  No checkpoint / tokenizer / enwik8 exists offline, so weights are seeded synthetic bf16 tensors and ids
are uniform random: throughput and roofline are meaningful, bits/byte is not (reported as null).

Contract line (rank 0): metric/value/unit/n_gpus/steps/warmup/ms_per_step/... + "roofline" (dominant kernel family,
HIP-event timed on the library's stream inside the timed region) + "cpu_baseline" (oracle, reference schedule, bounded
sample, rank 0 at N=1 only).
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


def make_streams(rank, n_streams, n_tokens, vocab):
    rng = np.random.default_rng(1000 + rank)
    return [np.concatenate([[0], rng.integers(0, vocab, n_tokens)]).astype(np.uint32) for _ in range(n_streams)]


def note(msg):
    """progress on stderr (long runs must not look hung to the GPU harness)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


# kernel-family name (cz_profile_read) -> rocprofv3 kernel names
FAMILY_KERNELS = {
    "dec_attention": ["dec_attn_kernel<3>"], "dec_gemm_plain": ["dec_gemm_kernel<0, 0, 1>"], "dec_gemm_qkv_rope": ["dec_gemm_kernel<1, 1, 1>"],
    "dec_gemm_swiglu": ["dec_gemm_kernel<2, 0, 2>"], "dec_gemm_lm_head": ["dec_gemm_kernel<0, 0, 8>"], "dec_rmsnorm": ["dec_rmsnorm_kernel"],
    "decode_symbol": ["cdf_par_kernel<1>"], "cdf_pairs": ["cdf_par_kernel<0>"],
    "gemm_bf16w_f32_mfma": ["gemm_bf16w_kernel<128, 128, 2, 2>"], "gemm_bf16w_f32_mfma_n64": ["gemm_bf16w_kernel<128, 64, 2, 2>"],
    "attention_f32_mfma": ["attention_v2_kernel<3, 16>", "attention_v2_kernel<3, 8>"],
}


def pmc_traffic(family, tag="pmc"):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (profiles/r*_pmc_*_pmc.json, made by
    tools/prof_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same bench in eager mode).  FETCH_SIZE / WRITE_SIZE
    are in KiB; on gfx950 FETCH_SIZE counts 128-B requests of wide coalesced reads at 64 B, so it is doubled (MI355X_MICROARCH.md,
    "HBM").  Returns (bytes_per_launch or None, note)."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    tot = 0.0; notes = []
    for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
        files = sorted(glob.glob(os.path.join(here, "profiles", f"r*_{tag}_{counter}_pmc.json")))
        if not files:
            notes.append(f"{counter}: no pass committed")
            continue
        d = json.load(open(files[-1]))
        disp = sum(d[k][counter]["dispatches"] for k in FAMILY_KERNELS.get(family, []) if k in d and counter in d[k])
        val = sum(d[k][counter]["sum"] for k in FAMILY_KERNELS.get(family, []) if k in d and counter in d[k])
        if not disp:
            notes.append(f"{counter}: kernel not in {os.path.basename(files[-1])}")
            continue
        tot += val * scale / disp
        notes.append(f"{counter} from {os.path.basename(files[-1])} ({disp} dispatches{', x2 gfx950 correction' if counter == 'FETCH_SIZE' else ''})")
    return (round(tot) if tot else None), "; ".join(notes)


def self_test_step(cz, sess, streams, context, reprime):
    """compress + decompress of every stream; returns (tokens, payload_bytes, t_encode, t_decode)"""
    t0 = time.perf_counter()
    pairs = sess.encode_tokens_batch(streams, context, reprime)
    payloads = []
    for lo, hi in pairs:
        p, zw = cz.ac_encode_pairs(lo, hi)
        payloads.append(p)
    t1 = time.perf_counter()
    outs = sess.decode_tokens_batch(payloads, [s.size - 1 for s in streams], context, reprime, 0)
    t2 = time.perf_counter()
    for s, o in zip(streams, outs):
        if not np.array_equal(s, o):
            raise RuntimeError("self-test round trip mismatch")
    note(f"self-test pass: {sum(s.size - 1 for s in streams)} tokens, compress {t1 - t0:.1f}s, decompress {t2 - t1:.1f}s, round trip exact")
    return sum(s.size - 1 for s in streams), sum(len(p) for p in payloads), t1 - t0, t2 - t1


def self_test_pipelined(cz, enc_sess, dec_sessions, streams, group, context, reprime):
    """The same work as self_test_step, scheduled as a pipeline over groups of `group` streams: the compute-bound compress of group
    k + 1 (big-M MFMA GEMMs) runs while the latency-bound lock-step decompress of groups <= k is in flight on the decode sessions
    (one per lane: own HIP stream, KV cache and step graph).  Returns (tokens, payload_bytes, compress busy s, decompress busy s)."""
    import threading
    groups = [streams[i:i + group] for i in range(0, len(streams), group)]
    L = len(dec_sessions)
    lane_thread = [None] * L
    err = []; stats = {"bytes": 0, "dec_busy": 0.0}
    lock = threading.Lock()

    def dec_worker(lane, grp, payloads):
        try:
            t = time.perf_counter()
            outs = dec_sessions[lane].decode_tokens_batch(payloads, [s.size - 1 for s in grp], context, reprime, 0)
            dt = time.perf_counter() - t
            for s, o in zip(grp, outs):
                if not np.array_equal(s, o):
                    raise RuntimeError("self-test round trip mismatch")
            with lock:
                stats["dec_busy"] += dt
        except Exception as e:      # noqa
            err.append(e)

    t0 = time.perf_counter(); enc_busy = 0.0
    for g, grp in enumerate(groups):
        t = time.perf_counter()
        pairs = enc_sess.encode_tokens_batch(grp, context, reprime)
        payloads = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
        enc_busy += time.perf_counter() - t
        stats["bytes"] += sum(len(p) for p in payloads)
        lane = g % L
        if lane_thread[lane] is not None:
            lane_thread[lane].join()
        lane_thread[lane] = threading.Thread(target=dec_worker, args=(lane, grp, payloads))
        lane_thread[lane].start()
    for t in lane_thread:
        if t is not None:
            t.join()
    if err:
        raise err[0]
    wall = time.perf_counter() - t0
    n = sum(s.size - 1 for s in streams)
    note(f"self-test pass (pipelined, {len(groups)} groups on {L} decode lanes): {n} tokens in {wall:.1f}s; compress busy {enc_busy:.1f}s, "
         f"decompress busy {stats['dec_busy']:.1f}s (lane-seconds), round trip exact")
    return n, stats["bytes"], enc_busy, stats["dec_busy"]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=384, help="streams (64 KiB chunks) per GPU, decoded in lock-step")
    ap.add_argument("--lanes", type=int, default=3, help="concurrent lock-step decode groups per GPU (library decode lanes)")
    ap.add_argument("--pipeline", type=int, default=0, help="1: compress group k+1 while groups <= k decompress (one decode session per lane)")
    ap.add_argument("--no-graph", action="store_true", help="launch the decode step eagerly (needed under rocprofv3 --pmc)")
    ap.add_argument("--kv-ctx-cap", type=int, default=0, help="tuning: positions per stream in the KV cache (0 = library default 2048)")
    ap.add_argument("--dec-nw-cap", type=int, default=0, help="tuning: waves per decode GEMM block (0 = library default 12)")
    ap.add_argument("--dec-grid-target", type=int, default=0, help="tuning: blocks per decode GEMM launch (0 = library default)")
    ap.add_argument("--lane-streams", type=int, default=128, help="streams per lock-step group")
    ap.add_argument("--warmup-tokens", type=int, default=1100, help="tokens/stream of a warmup step")
    ap.add_argument("--profile-tokens", type=int, default=1536, help="tokens/stream of the untimed per-family profiling pass")
    ap.add_argument("--tokens", type=int, default=0,
                    help="tokens per stream and step; default: one 64 KiB enwik8 sample (17408 tokens) spread over the K steps, "
                         "i.e. max(2048, 17408 / K rounded up to 512), so that the whole run stays within minutes for any K")
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--reprime-interval", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=640, help="tokens of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check only)")
    args = ap.parse_args()
    if args.tokens <= 0:
        args.tokens = 17408 if args.steps <= 1 else max(2048, ((17408 + args.steps - 1) // args.steps + 511) // 512 * 512)

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
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
    cfg = cz.tiny_llama_config() if args.tiny else cz.smollm2_135m_config()
    sess = cz.Session("smollm", cfg, seed=0x5EED, embed_std=0.02, device=local, max_streams=min(256, args.streams, args.lane_streams))
    sess.debug_set("max_lanes", args.lanes)
    if args.no_graph:
        sess.debug_set("no_graph", 1)
    if args.kv_ctx_cap:
        sess.debug_set("kv_ctx_cap", args.kv_ctx_cap)
    if args.dec_nw_cap:
        sess.debug_set("dec_nw_cap", args.dec_nw_cap)
    if args.dec_grid_target:
        sess.debug_set("dec_grid_target", args.dec_grid_target)
    dec_sessions = []
    if args.pipeline:
        nl = min(args.lanes, -(-args.streams // min(256, args.streams, args.lane_streams)))
        dec_sessions = [sess.fork() for _ in range(nl)]        # same device weights, own stream / KV cache / step graph

    def run_self_test(strs):
        if args.pipeline:
            return self_test_pipelined(cz, sess, dec_sessions, strs, min(256, args.streams, args.lane_streams), args.context, args.reprime_interval)
        return self_test_step(cz, sess, strs, args.context, args.reprime_interval)
    streams = make_streams(rank, args.streams, args.tokens, cfg.vocab)   # files shard across ranks: no data-path collective
    note(f"rank {rank}/{world}: session open, {args.streams} streams x {args.tokens} tokens")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warmup steps: the same streams cut to `--warmup-tokens` tokens — enough to size every buffer for the full
    # batch, build the decode hipGraph and page the code in; nothing is profiled
    sess.profile_enable(False)
    warm = [s[: args.warmup_tokens + 1] for s in streams]
    for _ in range(args.warmup):
        run_self_test(warm)

    # timed region: exactly K steps.  The big-M MFMA GEMM family (launched eagerly) is HIP-event timed on the library's
    # stream inside this region; the decode-step kernels run inside hipGraphs and cannot be bracketed one by one.
    sess.profile_enable(True, only=["gemm_bf16w_f32_mfma", "gemm_bf16w_f32_mfma_n64"])
    barrier()
    t0 = time.perf_counter()
    tok = byt = 0; te = td = 0.0
    for _ in range(args.steps):
        n, b, e, d = run_self_test(streams)
        tok += n; byt += b; te += e; td += d
    barrier()
    elapsed = time.perf_counter() - t0
    prof_timed = sess.profile_read()

    # per-family pass (untimed, eager, every family HIP-event timed) on the same streams cut to `--profile-tokens` tokens:
    # names the dominant kernel family and measures its achieved roofline
    # (one lock-step group alone on the device, so that the per-launch durations are not inflated by the other decode lanes)
    sess.profile_enable(True)
    short = [s[: args.profile_tokens + 1] for s in streams[: min(args.streams, args.lane_streams)]]
    if args.profile_tokens > 0:
        self_test_step(cz, sess, short, args.context, args.reprime_interval)
    prof_all = sess.profile_read()
    sess.profile_enable(False)
    dominant = max(prof_all, key=lambda k: prof_all[k]["total_ms"]) if prof_all else "gemm_bf16w_f32_mfma"
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
        def roof_of(name, st):
            if name.startswith(("gemm_bf16w", "attention_f32")):
                ach = st["flops"] / (st["total_ms"] * 1e-3) / 1e12 if st["total_ms"] else 0.0
                r = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4)}
            else:
                ach = st["bytes"] / (st["total_ms"] * 1e-3) / 1e9 if st["total_ms"] else 0.0
                r = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
            traffic, tnote = pmc_traffic(name, "pmc1536" if args.profile_tokens == 1536 else "pmc")
            r.update({"traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": tnote,
                      "algorithmic_per_launch": round((st["flops"] if r["bound"] == "mfma" else st["bytes"]) / max(st["launches"], 1)), "kernel": name, "launches": st["launches"], "avg_launch_us": round(1e3 * st["total_ms"] / max(st["launches"], 1), 3)})
            return r
        roof = roof_of(dominant, prof_all.get(dominant, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0}))
        roof["measured_in"] = (f"eager per-family pass inside bench.py ({len(short)} streams x {args.profile_tokens} tokens: the workload's own context "
                               "sawtooth, up to 1023 keys; one decode lane); the timed region replays hipGraphs on concurrent lanes")
        roof["families_ms_profile_pass"] = {k: round(v["total_ms"], 2) for k, v in prof_all.items()}
        zero = {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0}
        both = {k: prof_timed.get("gemm_bf16w_f32_mfma", zero)[k] + prof_timed.get("gemm_bf16w_f32_mfma_n64", zero)[k] for k in zero}
        roof_gemm = roof_of("gemm_bf16w_f32_mfma", both)
        roof_gemm["kernel"] = "gemm_bf16w_kernel<128,128> + <128,64> (both tile shapes of the big-M GEMM)"
        roof_gemm["traffic"] = None; roof_gemm["traffic_source"] = "see roofline (per-kernel PMC numbers are in profiles/)"
        roof_gemm["measured_in"] = "HIP events over the timed region"
        line = {
            "metric": "tokens/sec, enwik8 self-test (compress+decompress), smollm",
            "value": round(total_tok / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"enwik8 64 KiB samples, chunk-sharded: {args.streams} streams x {args.tokens} tokens per GPU and step"
                                   f"{'' if args.tokens == 17408 else ' (a 17408-token sample spread over the steps)'}, "
                                   f"SmolLM2-135M shape ({'tiny test model' if args.tiny else 'd576 L30 9q/3kv ffn1536 V49152'}), "
                                   f"context {args.context}, reprime_interval {args.reprime_interval}, seeded synthetic weights",
                       "streams_per_gpu": args.streams, "decode_lanes": min(args.lanes, -(-args.streams // min(256, args.streams, args.lane_streams))),
                       "streams_per_lane": min(256, args.streams, args.lane_streams), "tokens_per_stream": args.tokens, "parallelism": f"files x{world} (no data-path collective)"},
            "bits_per_byte": None,
            "bits_per_token_synthetic": round(8.0 * byt / max(tok, 1), 3),
            "compress_tokens_per_s_rank0": round(tok / te, 1) if te else None,
            "decompress_tokens_per_s_rank0": round(tok / td, 1) if td else None,
            "roofline": roof,
            "roofline_mfma_gemm": roof_gemm,
        }
        if world == 1 and args.cpu_sample > 0 and not args.tiny:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.context, args.reprime_interval)
        print(json.dumps(line), flush=True)
    for d_ in dec_sessions:
        d_.close()
    sess.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

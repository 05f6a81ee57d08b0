"""`candlezip` command line — the reference's CLI surface for the baseline (agent-off) path (src/main.rs:34-141, 143-221).

Global options come BEFORE the subcommand, like the reference's clap struct:
    python -m candlezip_amd [--backend smollm|rwkv7] [--model P ...] [--tokenizer P] [--context N] [--reprime-interval N]
                            {compress IN OUT | decompress IN OUT | self-test IN}
Same flag names and defaults; agent / scan flags are accepted and rejected ("not supported in this build") instead of
being silently ignored.  `--cpu` is rejected: this build has no CPU path.  Extras for machines without checkpoints
(none exist offline): `--synthetic SEED` (seeded weights) and `--token-ids FILE` (u32 ids instead of a tokenizer).

The heavy lifting (session, token loops, coder, container, hashes) is libcandlezip_hip.so through the C ABI; both
tokenizers are the library's own: SmolLM2's HF byte-level BPE tokenizer.json (cz_bpe_*, replacing the `tokenizers` crate calls at
src/main.rs:1812, 1852, 2654) and the RWKV World tokenizer (cz_rwkv_*).
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

import candlezip_amd as cz
from candlezip_amd import watchdog

_AGENT_FLAGS = ["--scan", "--agent", "--scan-attribution", "--scan-verbose"]
_AGENT_OPTS = ["--scan-chunk-size", "--scan-max-steps", "--scan-max-hint-tokens", "--scan-mcp-config", "--scan-output-dir", "--scan-python",
               "--scan-lookahead", "--scan-agent-timeout", "--scan-gate-threshold-pct", "--scan-gate-threshold-abs-bits", "--scan-agent-script",
               "--reuse-scan-dir", "--policy"]


def build_parser():
    p = argparse.ArgumentParser(prog="candlezip", description="Unified model-based compressor (MI355X hot path; agent layer not built)")
    p.add_argument("--backend", default="smollm")                                    # src/main.rs:38-39
    p.add_argument("--cpu", action="store_true")                                     # :42-43
    p.add_argument("--model")                                                        # :46-47
    p.add_argument("--model-repo", default="HuggingFaceTB/SmolLM2-135M")
    p.add_argument("--model-file", default="auto")
    p.add_argument("--tokenizer")
    p.add_argument("--tokenizer-repo", default="HuggingFaceTB/SmolLM2-135M")
    p.add_argument("--tokenizer-file", default="tokenizer.json")
    p.add_argument("--rwkv-model", default="modeldir/prepared/model.safetensors")    # :60-65
    p.add_argument("--rwkv-config", default="modeldir/prepared/config.json")
    p.add_argument("--rwkv-tokenizer", default="modeldir/rwkv_vocab_v20230424.json")
    p.add_argument("--context", type=int, default=512)                               # :68-69
    p.add_argument("--reprime-interval", type=int, default=512)                      # :71-72
    p.add_argument("--force-mismatch", action="store_true")                          # :122-123
    p.add_argument("--round-weights-to-bf16", action="store_true",
                   help="accept an F32 checkpoint whose values bf16 cannot hold (converted RWKV7 weights) by rounding them: a slightly "
                        "different model than the reference's f32 one; without this flag such a checkpoint is refused")
    p.add_argument("--reuse", action="store_true")                                   # :126-127 (agent cache; no effect without the agent)
    for f in _AGENT_FLAGS:
        p.add_argument(f, action="store_true")
    for o in _AGENT_OPTS:
        p.add_argument(o)
    # extras (not in the reference)
    p.add_argument("--synthetic", type=lambda s: int(s, 0), help="seeded synthetic weights instead of --model (no checkpoint exists offline)")
    p.add_argument("--config", help="config.json of the model (defaults to the one next to --model)")
    p.add_argument("--token-ids", help="u32 little-endian token ids (without BOS) instead of running a tokenizer on IN")
    p.add_argument("--device", type=int, default=0)
    sub = p.add_subparsers(dest="command", required=True)
    for name in ("compress", "decompress"):
        sp = sub.add_parser(name)
        sp.add_argument("input"); sp.add_argument("output")
    sp = sub.add_parser("self-test")
    sp.add_argument("input")
    return p


def _reject_agent(args):
    used = [f for f in _AGENT_FLAGS if getattr(args, f[2:].replace("-", "_"))] + [o for o in _AGENT_OPTS if getattr(args, o[2:].replace("-", "_")) is not None]
    if used:
        sys.exit(f"error: {', '.join(used)}: agentic conditioning / scan is not supported in this build (baseline entropy-coding path only)")
    if args.cpu:
        sys.exit("error: --cpu: this build has no CPU path (MI355X only)")


def _hash_files(paths):
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    out = np.zeros(16, np.uint8)
    cz._check(cz.lib().cz_blake3_files_bin16(arr, len(paths), out.ctypes.data))
    return out.tobytes()


def _hash_file(path):
    out = np.zeros(16, np.uint8)
    cz._check(cz.lib().cz_blake3_file_bin16(path.encode(), out.ctypes.data))
    return out.tobytes()


class _HfTokenizer:
    """the HF library behind the interface of cz.BpeTokenizer (only for tokenizer.json layouts the in-process tokenizer refuses)"""

    def __init__(self, tok):
        self.tok = tok

    def vocab_size(self):
        return self.tok.get_vocab_size(True)

    def encode(self, data: bytes):
        return np.array(self.tok.encode(data.decode("utf-8", "replace"), add_special_tokens=False).ids, np.uint32)

    def decode(self, ids, skip_special_tokens=True) -> bytes:
        return self.tok.decode([int(i) for i in ids], skip_special_tokens=skip_special_tokens).encode("utf-8")


class _Model:
    """weights + tokenizer + hashes for one backend (what encode_file / decode_file set up at src/main.rs:1799-1829, 2444-2466)"""

    def __init__(self, args, vocab_from_header=None):
        self.backend = args.backend.lower()
        if self.backend not in ("smollm", "rwkv7"):
            sys.exit(f"unknown backend {self.backend}")                                # :1829
        self.bos = 0
        self.tok = None
        if self.backend == "smollm":
            if args.synthetic is not None:
                cfg = cz.smollm2_135m_config()
                self.sess = cz.Session("smollm", cfg, seed=args.synthetic, device=args.device, max_streams=1)
                self.model_hash = cz.blake3_16(f"synthetic-smollm2-135m:{args.synthetic}".encode())
                self.repr = b"synthetic"
            else:
                if not args.model:
                    sys.exit("error: --model PATH (safetensors) is required: no network, the HF hub is not reachable from this build")
                paths = [args.model] if os.path.isfile(args.model) else sorted(
                    os.path.join(args.model, f) for f in os.listdir(args.model) if f.endswith(".safetensors"))
                cfgp = args.config or os.path.join(os.path.dirname(paths[0]), "config.json")
                self.sess = cz.Session("smollm", None, weight_paths=paths, config_path=cfgp, device=args.device, max_streams=1,
                                       round_f32_weights_to_bf16=args.round_weights_to_bf16)
                self.model_hash = _hash_files(paths)                                   # blake3_files_bin16 (:1810)
                self.repr = os.path.basename(paths[0]).encode() if len(paths) == 1 else b"model.safetensors.index.json"
            if args.tokenizer:
                try:
                    self.tok = cz.BpeTokenizer(args.tokenizer)                         # in-process byte-level BPE (csrc/cz_bpe.cpp), :1812
                except cz.CandlezipError as e:
                    if "CZ_ERR_UNSUPPORTED" not in str(e):
                        raise
                    from tokenizers import Tokenizer                                   # a layout the in-process tokenizer refuses: the HF library itself
                    print(f"note: {e}; using the HF tokenizers library", file=sys.stderr)
                    self.tok = _HfTokenizer(Tokenizer.from_file(args.tokenizer))
                self.tok_hash = _hash_file(args.tokenizer)                             # :1811
            else:
                self.tok_hash = bytes(16)
            self.vocab = self.sess.vocab_size() if self.tok is None else self.tok.vocab_size()   # get_vocab_size(true), :1814
        else:
            if args.synthetic is not None:
                self.sess = cz.Session("rwkv7", cz.rwkv7_0b1_config(), seed=args.synthetic, device=args.device, max_streams=1)
                self.model_hash = cz.blake3_16(f"synthetic-rwkv7-0.1b:{args.synthetic}".encode())
                self.repr = b"synthetic"
            else:
                # Rwkv7Session::load(model_path, config_path, device) (src/models.rs:132-146): the config file decides the shape
                self.sess = cz.Session("rwkv7", None, weight_paths=[args.rwkv_model], config_path=args.rwkv_config, device=args.device, max_streams=1,
                                       round_f32_weights_to_bf16=args.round_weights_to_bf16)
                self.model_hash = _hash_file(args.rwkv_model)                          # :1822
                self.repr = os.path.basename(args.rwkv_model).encode()
            self.vocab = self.sess.vocab_size()
            self.rwkv_tok = None
            if os.path.exists(args.rwkv_tokenizer):
                self.rwkv_tok = cz.WorldTokenizer(args.rwkv_tokenizer)             # Tokenizer::new (rwkv7.rs:553), :1823
                self.tok_hash = _hash_file(args.rwkv_tokenizer)
            else:
                self.tok_hash = bytes(16)

    def tokenize(self, data: bytes, args):
        if args.token_ids:
            return np.fromfile(args.token_ids, "<u4").astype(np.uint32)
        if self.backend == "smollm":
            if self.tok is None:
                sys.exit("error: --tokenizer tokenizer.json (or --token-ids) is required for --backend smollm")
            if data.decode("utf-8", "replace").encode("utf-8") != data:               # String::from_utf8_lossy (:1852)
                print("warning: input is not valid UTF-8; like the reference, the round trip will be lossy", file=sys.stderr)
            return np.asarray(self.tok.encode(data), np.uint32)
        if self.rwkv_tok is None:                                                      # complete fallback: literals (:863-868)
            return (np.frombuffer(data, np.uint8).astype(np.uint32) + self.vocab)
        return self.rwkv_tok.plan_symbols(data, self.vocab)

    def detokenize(self, ids, args):
        if self.backend == "smollm":
            if self.tok is None:
                sys.exit("error: --tokenizer is required to turn ids back into text (or decode with --token-ids to get ids)")
            return self.tok.decode(ids, True)                                          # decode(ids, skip_special_tokens = true), :2654
        if self.rwkv_tok is None:
            return bytes(int(i) - self.vocab for i in ids if int(i) >= self.vocab)
        return self.rwkv_tok.detok(ids, self.vocab)


def _check_window(args):
    """the reference's encoder keeps min(context, 511) tokens at a re-prime, its decoder min(context - 1, 511) (src/main.rs:1935 vs :2506):
    for context < 512 the two sides disagree and the archive does not decode — reproduced literally, so say it loudly"""
    if args.backend.lower() != "smollm":
        return
    if args.context <= 0:
        sys.exit("error: --context must be > 0 (the reference fails with 'reprime called with empty history')")
    if args.reprime_interval <= 0:
        sys.exit("error: --reprime-interval must be > 0")
    if args.context < 512:
        print(f"warning: --context {args.context} < 512: encoder and decoder disagree on the re-prime window (min({args.context}, 511) vs "
              f"min({args.context} - 1, 511), src/main.rs:1935 / :2506); like the reference's, an archive longer than one re-prime interval "
              "will NOT decode", file=sys.stderr)


def _dist(args):
    """one process per GPU (torchrun): returns (torch.distributed or None, rank).  RCCL ("nccl") on a multi-GPU node; CZ_DIST_BACKEND=gloo
    rehearses the same path on one GPU / in CPU tests.  Nothing in the token loop crosses ranks."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CZ_DIST_BACKEND", "nccl")
        if backend == "nccl":
            args.device = local
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank


def compress(args, inp, outp, model=None):
    t0 = time.perf_counter()
    _check_window(args)
    dist, rank = _dist(args)
    m = model or _Model(args)
    data = open(inp, "rb").read()
    ids = np.concatenate([[m.bos], m.tokenize(data, args)]).astype(np.uint32)          # :1850-1866
    if dist is not None and m.backend == "smollm":
        # ONE file over the node's GPUs: every rank evaluates a block of whole re-prime segments, rank 0 gathers the (c_lo, c_hi) pairs,
        # runs the serial coder and writes the single container (src/main.rs:2358-2376)
        from candlezip_amd.parallel import encode_sharded
        lo, hi = encode_sharded(lambda b, e: m.sess.encode_tokens_range_batch([ids], [(b, e)], args.context, args.reprime_interval)[0],
                                ids.size - 1, args.reprime_interval, dist)
        if rank != 0:
            return m
    elif dist is not None and rank != 0:
        return m                                                                       # RWKV7 has no re-prime: one stream does not shard
    else:
        lo, hi = m.sess.encode_tokens(ids, args.context, args.reprime_interval)
    m.trace = {"encode_ids": ids[1:].copy(), "c_lo": lo, "c_hi": hi}                   # watchdog-lite (watchdog.py)
    payload, zero_width = cz.ac_encode_pairs(lo, hi)
    if zero_width:
        print(f"warning: {zero_width} token(s) have a zero-width coding interval (p < 2^-30): like the reference's, this file will NOT decode", file=sys.stderr)
    h = cz.Header()
    h.bos_token_id, h.token_count, h.orig_len_bytes = m.bos, ids.size - 1, len(data)
    for dst, src in ((h.model_hash16, m.model_hash), (h.tokenizer_hash16, m.tok_hash), (h.orig_hash16, cz.blake3_16(data))):
        C.memmove(dst, src, 16)
    chunk = 2048 if m.backend == "rwkv7" else args.reprime_interval                    # agent_chunk (:1869), packed even when the agent is off
    h.reserved_flags = cz.lib().cz_flags_pack(0, 0, 0, chunk) | cz.FLAG_PRODUCER_HIP    # producer mark (include/candlezip.h)
    h.context_window, h.vocab_size, h.reprime_interval = args.context, m.vocab, args.reprime_interval
    blob = cz.write_header(h, m.repr) + payload                                        # :2362-2376
    open(outp, "wb").write(blob)
    el = time.perf_counter() - t0
    chars = len(data.decode("utf-8", "replace"))
    bpb = 8.0 * len(blob) / max(len(data), 1)
    print(f"Encoded: {len(data)} bytes -> {len(blob)} bytes | bits/byte={bpb:.3f} | bits/char={8.0 * len(blob) / max(chars, 1):.3f} | context={args.context} | time={el:.2f}s")
    return m


def decompress(args, inp, outp, model=None):
    dist, rank = _dist(args)
    if dist is not None and rank != 0:
        return model                                                                   # one stream decodes sequentially: rank 0 only
    buf = open(inp, "rb").read()
    h, repr_, n = cz.read_header(buf)                                                  # :2411
    if h.reserved_flags & 1:
        sys.exit("file requires --agent for deterministic decoding (agentic conditioning is not supported in this build)")   # :2421
    if h.reserved_flags & 4:
        used = C.c_size_t(); a = np.frombuffer(buf[n:], np.uint8)
        cz._check(cz.lib().cz_skip_gate_section(a.ctypes.data, a.size, C.byref(used)))
        n += used.value
    m = model or _Model(args)
    if not (h.reserved_flags & cz.FLAG_PRODUCER_HIP):
        msg = ("this .canz does not carry the producer mark of this build (reserved_flags bit 15): it was written by another implementation "
               "(e.g. the candle reference), whose logits arithmetic differs, so its payload would decode to garbage here")
        if not args.force_mismatch:
            sys.exit(f"producer mismatch: {msg}. Use --force-mismatch to override.")
        print(f"warning: {msg}", file=sys.stderr)
    if not args.force_mismatch:                                                        # :2448-2459
        if m.model_hash != bytes(h.model_hash16):
            sys.exit("model hash mismatch. Use --force-mismatch to override.")
        if m.tok_hash != bytes(h.tokenizer_hash16):
            sys.exit("tokenizer hash mismatch. Use --force-mismatch to override.")
    ids = m.sess.decode_tokens(buf[n:], int(h.token_count), h.context_window, h.reprime_interval, h.bos_token_id)
    if getattr(m, "trace", None) is not None:
        m.trace["decode_ids"] = ids[1:].copy()
    if args.token_ids and outp.endswith(".ids"):
        ids[1:].astype("<u4").tofile(outp)
    else:
        open(outp, "wb").write(m.detokenize(ids[1:], args))
    return m


def self_test(args, inp):
    stem = os.path.splitext(inp)[0]
    out, dec = stem + ".canz", stem + ".roundtrip.txt"                                 # :157-158
    t0 = time.perf_counter()
    m = compress(args, inp, out)
    t1 = time.perf_counter()
    decompress(args, out, dec, m)
    t2 = time.perf_counter()
    src, enc, got = open(inp, "rb").read(), open(out, "rb").read(), open(dec, "rb").read()
    print(f"Compression   : {t1 - t0:.2f}s")                                            # :191-193
    print(f"Decompression : {t2 - t1:.2f}s")
    print(f"Bits per byte : {8.0 * len(enc) / max(len(src), 1):.3f}")
    if watchdog.enabled() and getattr(m, "trace", None) and "decode_ids" in m.trace:
        d = watchdog.run_dir(inp, args.backend)
        watchdog.write_steps(d, "watchdog_encode_steps.jsonl", m.trace["encode_ids"], "encode")
        watchdog.write_steps(d, "watchdog_decode_steps.jsonl", m.trace["decode_ids"], "decode")
    if src == got:
        print(f"Roundtrip OK. Encoded: {len(enc)} bytes, Decoded: {len(got)} bytes")
        return 0
    print("[watchdog] Roundtrip mismatch detected. Collecting diagnostics...", file=sys.stderr)       # :195
    tr = getattr(m, "trace", None) or {}
    if "encode_ids" in tr and "decode_ids" in tr:
        d = watchdog.run_dir(inp, args.backend)
        watchdog.write_steps(d, "watchdog_encode_steps.jsonl", tr["encode_ids"], "encode")
        watchdog.write_steps(d, "watchdog_decode_steps.jsonl", tr["decode_ids"], "decode")
        rec = watchdog.report(d, tr["encode_ids"], tr["decode_ids"], tr.get("c_lo"), tr.get("c_hi"))
        if rec:
            print(f"[watchdog] first differing symbol at i={rec['i']}: encode_sym={rec['encode_sym']} decode_sym={rec['decode_sym']}"
                  f" (zero-width coding intervals: {rec.get('zero_width_total', 0)}, first at {rec.get('zero_width_steps', [])[:4]}); report in {d}",
                  file=sys.stderr)
        else:
            print("[watchdog] token ids round-trip exactly: the difference is in (de)tokenisation of the text", file=sys.stderr)
    return 1


def main(argv=None):
    args = build_parser().parse_args(argv)
    _reject_agent(args)
    rc = 0
    if args.command == "compress":
        compress(args, args.input, args.output)
    elif args.command == "decompress":
        decompress(args, args.input, args.output)
    else:
        dist, rank = _dist(args)
        if dist is not None and rank != 0:                                             # ranks > 0 only help with the compress half
            compress(args, args.input, os.devnull)
        else:
            rc = self_test(args, args.input)
    dist, _ = _dist(args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()

"""Corpus self-test: BASELINE configs 3 and 5 on real files — split every file into chunks (64 KiB by default; a cut may fall inside a
UTF-8 sequence, which makes that chunk's round trip lossy exactly as the reference's `from_utf8_lossy` would), tokenize each chunk,
shard the chunks over the ranks (LPT by token count, no data-path collective), and on every rank compress + decompress its share in
lock-step batches.  Reports tokens/s and bits/byte (meaningful only with real weights) and verifies every chunk's round trip.

    python -m candlezip_amd.corpus --tokenizer tok.json [--model DIR | --synthetic SEED] [--chunk-bytes 65536] FILE...
    python -m candlezip_amd.corpus --backend rwkv7 [--rwkv-tokenizer vocab.json] [--model F --rwkv-config C | --synthetic SEED] FILE...   (config 4)
    torchrun --nproc-per-node N -m candlezip_amd.corpus ...            (one rank per GPU; RANK/LOCAL_RANK/WORLD_SIZE from the env)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import candlezip_amd as cz
from candlezip_amd.parallel import run_sharded


def chunk_file(path, chunk_bytes):
    data = open(path, "rb").read()
    return [(os.path.basename(path), o, data[o:o + chunk_bytes]) for o in range(0, max(len(data), 1), chunk_bytes)]


class _WorldTok:
    """the RWKV side of the same interface: symbols = plan_rwkv_symbols(bytes) (token ids, or vocab + byte literals), text back with
    rwkv_detok_with_literals (src/main.rs:833-895); without a vocabulary file every byte is a literal symbol (:863-868).  Byte-exact."""
    lossless = True

    def __init__(self, vocab_json, vocab):
        self.tok = cz.WorldTokenizer(vocab_json) if vocab_json else None
        self.vocab = vocab

    def encode(self, data: bytes):
        if self.tok is None:
            return np.frombuffer(data, np.uint8).astype(np.uint32) + self.vocab
        return self.tok.plan_symbols(data, self.vocab)

    def decode(self, ids, _skip_special=True) -> bytes:
        if self.tok is None:
            return bytes(int(i) - self.vocab for i in ids if int(i) >= self.vocab)
        return self.tok.detok(ids, self.vocab)

    def close(self):
        if self.tok is not None:
            self.tok.close()


def self_test_chunks(sess, tok, chunks, context=512, reprime=512, bos=0):
    """chunks: list of bytes.  Returns per chunk (n_tokens, payload_bytes, roundtrip_ok, lossless, BLAKE3-128 of the payload)"""
    ids = [np.concatenate([[bos], tok.encode(c)]).astype(np.uint32) for c in chunks]
    pairs = sess.encode_tokens_batch(ids, context, reprime)
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    outs = sess.decode_tokens_batch(pays, [i.size - 1 for i in ids], context, reprime, bos)
    res = []
    for c, i, p, o in zip(chunks, ids, pays, outs):
        text = tok.decode(o[1:], True)
        lossy = c if getattr(tok, "lossless", False) else c.decode("utf-8", "replace").encode("utf-8")   # what the reference itself can give back (from_utf8_lossy)
        res.append((int(i.size - 1), len(p), bool(np.array_equal(i, o)) and text == lossy, lossy == c, cz.blake3_16(p).hex()))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--backend", default="smollm", choices=["smollm", "rwkv7"])        # src/main.rs:38-39
    ap.add_argument("--tokenizer", help="HF tokenizer.json (byte-level BPE); required for --backend smollm")
    ap.add_argument("--rwkv-tokenizer", help="RWKV World vocabulary json (src/main.rs:64-65); without it every byte is a literal symbol")
    ap.add_argument("--rwkv-config", help="config.json next to --model for --backend rwkv7 (src/main.rs:62-63)")
    ap.add_argument("--model", help="directory or file with safetensors weights (+ config.json)")
    ap.add_argument("--synthetic", type=int, help="seeded synthetic weights of the backend's BASELINE shape (SmolLM2-135M / RWKV7-0.1B) instead of --model")
    ap.add_argument("--chunk-bytes", type=int, default=65536)
    ap.add_argument("--context", type=int, default=512)
    ap.add_argument("--reprime-interval", type=int, default=512)
    ap.add_argument("--max-streams", type=int, default=128)
    args = ap.parse_args(argv)
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CZ_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            local = 0
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.synthetic is None and not args.model:
        ap.error("--model or --synthetic is required")
    if args.backend == "smollm":
        if not args.tokenizer:
            ap.error("--tokenizer is required for --backend smollm")
        tok = cz.BpeTokenizer(args.tokenizer)
        if args.synthetic is not None:
            sess = cz.Session("smollm", cz.smollm2_135m_config(), seed=args.synthetic, device=local, max_streams=args.max_streams)
        else:
            paths = [args.model] if os.path.isfile(args.model) else sorted(os.path.join(args.model, f) for f in os.listdir(args.model) if f.endswith(".safetensors"))
            sess = cz.Session("smollm", None, weight_paths=paths, config_path=os.path.join(os.path.dirname(paths[0]), "config.json"), device=local,
                              max_streams=args.max_streams)
    else:                                                                    # BASELINE config 4: the recurrent-state kernel path (no re-prime)
        if args.synthetic is not None:
            sess = cz.Session("rwkv7", cz.rwkv7_0b1_config(), seed=args.synthetic, device=local, max_streams=args.max_streams)
        else:
            sess = cz.Session("rwkv7", None, weight_paths=[args.model], config_path=args.rwkv_config or os.path.join(os.path.dirname(args.model), "config.json"),
                              device=local, max_streams=args.max_streams)
        tok = _WorldTok(args.rwkv_tokenizer, sess.vocab_size())
    chunks = [c for f in args.files for c in chunk_file(f, args.chunk_bytes)]
    sizes = [len(c[2]) for c in chunks]                                     # bytes as the LPT weight (token counts are proportional)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    res = run_sharded(chunks, sizes, lambda mine: self_test_chunks(sess, tok, [c[2] for c in mine], args.context, args.reprime_interval), dist)
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if rank == 0:
        ntok = sum(r[0] for r in res); nby = sum(sizes); canz = sum(r[1] for r in res)
        bad = [f"{c[0]}@{c[1]}" for c, r in zip(chunks, res) if not r[2]]
        print(json.dumps({"backend": args.backend, "chunks": len(chunks), "files": len(args.files), "ranks": world, "tokens": ntok, "input_bytes": nby, "payload_bytes": canz,
                          "self_test_tokens_per_s": round(ntok / wall, 1), "payload_bits_per_byte": round(8.0 * canz / max(nby, 1), 4),
                          "lossy_utf8_chunks": sum(1 for r in res if not r[3]), "roundtrip_failures": bad,
                          "payload_digest": cz.blake3_16("".join(r[4] for r in res).encode()).hex()}))     # over the chunks in file order: independent of the rank count
    tok.close(); sess.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0 if not (rank == 0 and bad) else 1


if __name__ == "__main__":
    sys.exit(main())

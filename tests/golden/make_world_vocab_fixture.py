#!/usr/bin/env python3
"""Fixture for the RWKV "World" tokenizer.  rwkv_vocab_v20230424.json does not exist offline, so a vocabulary of
the SAME KIND — a JSON object {token string: id}, which is what Tokenizer::new reads (candle_rwkv7/src/models/rwkv7.rs:553-556:
HashMap<String, u32>, keys taken as their UTF-8 bytes) — is derived here from the committed reference inputs: id 0 is left unused
(the World vocabulary keeps 0 for <|endoftext|>), ids 1..128 are the single ASCII bytes (JSON strings cannot express a lone byte
>= 0x80, so like the real file's JSON form the vocabulary has no such single-byte tokens), then every non-ASCII character of the
inputs, then the most frequent 2..8-byte ASCII n-grams, 511 tokens in all so that the ids also fit the 512-symbol tiny test models.
Output: world_vocab_test.json.   Run from the repo root:  python tests/golden/make_world_vocab_fixture.py"""
import json, os
from collections import Counter

here = os.path.dirname(os.path.abspath(__file__))
files = ["alice29.txt", "asyoulik.txt", "enwik8_128kb_0", "fields.c"]
raw = {n: open(os.path.join(here, "inputs", n), "rb").read() for n in files}
vocab = {}
nxt = 1
for b in range(128):
    vocab[chr(b)] = nxt; nxt += 1
chars = Counter(ch for n in files for ch in raw[n].decode("utf-8", "replace") if ord(ch) > 127 and ch != "�")
for ch, _ in sorted(chars.items(), key=lambda kv: (-kv[1], kv[0])):
    if nxt >= 300:
        break
    vocab[ch] = nxt; nxt += 1
grams = Counter()
for n in files:
    d = raw[n]
    for L in (2, 3, 4, 5, 6, 8):
        for i in range(0, len(d) - L, 1):
            g = d[i:i + L]
            if max(g) < 128:
                grams[g] += 1
# longer n-grams first in benefit order: (len - 1) * count
for g, _ in sorted(grams.items(), key=lambda kv: (-(len(kv[0]) - 1) * kv[1], kv[0])):
    if nxt >= 512:
        break
    s = g.decode("ascii")
    if s not in vocab:
        vocab[s] = nxt; nxt += 1
json.dump(vocab, open(os.path.join(here, "world_vocab_test.json"), "w"), ensure_ascii=True, indent=0, sort_keys=False)
print(len(vocab), "tokens, max id", max(vocab.values()))

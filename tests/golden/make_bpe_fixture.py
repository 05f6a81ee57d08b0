#!/usr/bin/env python3
"""Fixture for the in-process byte-level BPE tokenizer.  SmolLM2's tokenizer.json does not exist offline, so a
tokenizer of the SAME KIND (Sequence[Digits(individual_digits), ByteLevel(use_regex)] -> BPE -> ByteLevel decoder, two special
tokens) is trained here with the HF `tokenizers` library — the library the reference links (Cargo.toml:25, src/main.rs:1812,
1852, 2654) — on the committed reference inputs, and its own encodings/decodings of a set of hard cases are recorded.
Outputs: bpe_test_tokenizer.json, bpe_cases.json.   Run from the repo root:  python tests/golden/make_bpe_fixture.py"""
import hashlib, json, os
import tokenizers
from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers

here = os.path.dirname(os.path.abspath(__file__))
tok = Tokenizer(models.BPE())
tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Digits(individual_digits=True), pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)])
tok.decoder = decoders.ByteLevel()
trainer = trainers.BpeTrainer(vocab_size=3000, special_tokens=["<|endoftext|>", "<|im_start|>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
files = ["alice29.txt", "asyoulik.txt", "enwik8_128kb_0", "fields.c"]
raw = {n: open(os.path.join(here, "inputs", n), "rb").read() for n in files}
tok.train_from_iterator([raw[n].decode("utf-8", "replace") for n in files], trainer)
tok.save(os.path.join(here, "bpe_test_tokenizer.json"))

cases = [
    b"", b" ", b"  ", b"\n", b"a", b"Hello  world 1234 it's\n\n  x\t<|endoftext|> caf\xc3\xa9 \xe4\xb8\xad\xe6\x96\x87 \xf0\x9f\x98\x80",
    b"I'm we're they've she'll he'd it's don't 'S 'T  'll'll",             # contractions are case-sensitive
    b"trailing spaces   ", b"   leading", b"tabs\t\tand\r\nCRLF\r\n\r\n", b"a \n b", b"x  \n\n\n   y", b" \t \n",
    b"3.14159 1,000,000 v2.0 x86_64 0x1F 1e-9 \xc2\xb2 \xc2\xbd \xd9\xa1\xd9\xa2\xd9\xa3 \xe2\x85\xa7",    # superscript, fraction, Arabic-Indic digits, roman numeral
    b"!!! ... --- ((())) <<>> #$%^&* a.b,c;d:e", b"snake_case CamelCase kebab-case path/to/file.txt user@example.com",
    b"\xc3\x9cber stra\xc3\x9fe na\xc3\xafve \xd0\x9f\xd1\x80\xd0\xb8\xd0\xb2\xd0\xb5\xd1\x82 \xce\xb1\xce\xb2\xce\xb3 \xe6\x97\xa5\xe6\x9c\xac\xe8\xaa\x9e\xe3\x81\xae",
    b"\xe2\x80\x83em\xe2\x80\x83space \xc2\xa0nbsp\xc2\xa0 \xe3\x80\x80ideographic",                    # Unicode white space
    b"bad \xff\xfe utf8 \xc3( \xe2\x82 end \xf0\x9f\x98", b"\x80\x80", b"nul\x00byte",
    b"<|endoftext|><|im_start|>x<|endoftext|", b"a<|endoftext|>b", b"[[Wikipedia:Link|text]] {{template|1=2}} &amp; &lt;ref&gt;",
    raw["fields.c"][:4096], raw["alice29.txt"][1000:5096], raw["enwik8_128kb_0"][:4096], raw["asyoulik.txt"][-3000:],
]
out = {"tokenizers_version": tokenizers.__version__, "cases": [], "files": {}}
for c in cases:
    text = c.decode("utf-8", "replace")                                  # String::from_utf8_lossy (src/main.rs:1852)
    ids = tok.encode(text, add_special_tokens=False).ids
    out["cases"].append({"bytes_hex": c.hex(), "ids": ids, "decoded_hex": tok.decode(ids, skip_special_tokens=True).encode("utf-8").hex()})
for n in files:
    ids = tok.encode(raw[n].decode("utf-8", "replace"), add_special_tokens=False).ids
    out["files"][n] = {"count": len(ids), "sha256_le_u32": hashlib.sha256(b"".join(int(i).to_bytes(4, "little") for i in ids)).hexdigest(),
                       "roundtrip_equal": tok.decode(ids, skip_special_tokens=True).encode("utf-8") == raw[n].decode("utf-8", "replace").encode("utf-8")}
json.dump(out, open(os.path.join(here, "bpe_cases.json"), "w"))
print("cases", len(cases), {n: v["count"] for n, v in out["files"].items()})

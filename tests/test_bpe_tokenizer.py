"""In-process byte-level BPE against the HF `tokenizers` library — the library the reference links
(Cargo.toml:25; calls at src/main.rs:1812, 1852, 2654).  The fixture (tests/golden/make_bpe_fixture.py) holds a tokenizer of
SmolLM2's kind trained on the committed reference inputs plus the library's own encodings of hard cases; when the wheel is
importable the comparison is also made live on whole files.  No GPU needed: host code of libcandlezip_hip.so."""
import hashlib
import json
import os
import numpy as np
import pytest


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    m.lib()
    return m


@pytest.fixture(scope="module")
def bpe(cz, golden):
    t = cz.BpeTokenizer(os.path.join(golden, "bpe_test_tokenizer.json"))
    yield t
    t.close()


def test_cases_match_the_hf_library(cz, bpe, golden):
    fx = json.load(open(os.path.join(golden, "bpe_cases.json")))
    assert bpe.vocab_size() == 3000
    for k, c in enumerate(fx["cases"]):
        raw = bytes.fromhex(c["bytes_hex"])
        ids = bpe.encode(raw)
        assert ids.tolist() == c["ids"], (k, raw[:60])
        assert bpe.decode(ids, True) == bytes.fromhex(c["decoded_hex"]), (k, raw[:60])
    # special tokens are cut out of the raw text and dropped again by decode(skip_special_tokens=true)
    ids = bpe.encode(b"a<|endoftext|>b")
    assert 0 in ids.tolist() and bpe.decode(ids, True) == b"ab" and bpe.decode(ids, False) == b"a<|endoftext|>b"


def test_whole_files_match_and_roundtrip(cz, bpe, golden):
    fx = json.load(open(os.path.join(golden, "bpe_cases.json")))
    for name, want in fx["files"].items():
        raw = open(os.path.join(golden, "inputs", name), "rb").read()
        ids = bpe.encode(raw)
        assert ids.size == want["count"], name
        assert hashlib.sha256(ids.astype("<u4").tobytes()).hexdigest() == want["sha256_le_u32"], name
        assert bpe.decode(ids) == raw.decode("utf-8", "replace").encode("utf-8"), name     # from_utf8_lossy both ways


def test_live_against_tokenizers_wheel(cz, bpe, golden):
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(os.path.join(golden, "bpe_test_tokenizer.json"))
    rng = np.random.default_rng(3)
    raw = open(os.path.join(golden, "inputs", "enwik8_128kb_0"), "rb").read()
    samples = [raw[o:o + n] for o, n in zip(rng.integers(0, len(raw) - 3000, 40), rng.integers(1, 3000, 40))]   # cuts inside UTF-8 sequences too
    samples += [bytes(rng.integers(0, 256, 300, dtype=np.uint8)) for _ in range(10)]                             # binary garbage
    samples += [b" " * 5000 + b"x", b"=" * 7000, b"9" * 300, b"\n" * 100, ("é" * 2000).encode()]          # long single-class runs
    for s in samples:
        text = s.decode("utf-8", "replace")
        want = hf.encode(text, add_special_tokens=False).ids
        got = bpe.encode(s)
        assert got.tolist() == want, s[:40]
        assert bpe.decode(got) == hf.decode(want, skip_special_tokens=True).encode("utf-8")


def test_unsupported_layouts_are_refused(cz, golden, tmp_path):
    d = json.load(open(os.path.join(golden, "bpe_test_tokenizer.json")))
    for patch, msg in (({"normalizer": {"type": "NFC"}}, "normalizer"), ({"pre_tokenizer": {"type": "Whitespace"}}, "not supported"),
                       ({"decoder": {"type": "WordPiece"}}, "decoder")):
        e = dict(d); e.update(patch)
        p = tmp_path / "t.json"; p.write_text(json.dumps(e))
        with pytest.raises(cz.CandlezipError) as ei:
            cz.BpeTokenizer(str(p))
        assert "CZ_ERR_UNSUPPORTED" in str(ei.value) and msg in str(ei.value)
    with pytest.raises(cz.CandlezipError):
        cz.BpeTokenizer(str(tmp_path / "missing.json"))

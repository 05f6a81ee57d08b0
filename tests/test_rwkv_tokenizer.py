"""RWKV "World" tokenizer + symbol planning: cz_rwkv_tokenizer_open / cz_rwkv_plan_symbols / cz_rwkv_detok of
libcandlezip_hip.so against the oracle's restatement (oracle/czo.py: WorldTokenizer, plan_rwkv_symbols, rwkv_detok_with_literals,
following candle_rwkv7/src/models/rwkv7.rs:546-610 and src/main.rs:833-895).  The real rwkv_vocab_v20230424.json does not exist
offline; the fixture (tests/golden/make_world_vocab_fixture.py) is a vocabulary of the same kind derived from the committed
reference inputs.  The reference holds no vectors for these functions: the pins are the restatement, the greedy
longest-match property re-derived here by brute force, and byte-exact round trips.  No GPU needed: host code of the library."""
import json
import os
import numpy as np
import pytest

V = 512          # model vocabulary of the tiny test models; the fixture's ids are 1..511


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    m.lib()
    return m


@pytest.fixture(scope="module")
def vocab_path(golden):
    return os.path.join(golden, "world_vocab_test.json")


@pytest.fixture(scope="module")
def wt(cz, vocab_path):
    t = cz.WorldTokenizer(vocab_path)
    yield t
    t.close()


@pytest.fixture(scope="module")
def ot(czo, vocab_path):
    return czo.WorldTokenizer(vocab_path)


def _inputs(golden):
    for n in ("alice29.txt", "asyoulik.txt", "fields.c", "enwik8_128kb_0"):
        yield n, open(os.path.join(golden, "inputs", n), "rb").read()


def test_whole_files_match_the_oracle_and_round_trip(czo, wt, ot, golden):
    for name, data in _inputs(golden):
        ids = wt.plan_symbols(data, V)
        ref = czo.plan_rwkv_symbols(data, ot, V)
        assert ids.tolist() == ref, name
        assert ids.max() < V, name                                   # the fixture covers these files: no literal symbols
        assert wt.detok(ids, V) == data == czo.rwkv_detok_with_literals(ot, ref, V), name


def test_greedy_choice_is_the_highest_id_prefix_match(wt, vocab_path, golden):
    """encode_bytes takes, at every position, the FIRST table entry that is a prefix of the rest — table rows are in descending id
    order — and falls back to the single byte (rwkv7.rs:589-597).  Re-derived here by brute force over the raw vocabulary."""
    vocab = {k.encode(): v for k, v in json.load(open(vocab_path)).items()}
    data = open(os.path.join(golden, "inputs", "fields.c"), "rb").read()[:3000]
    ids = wt.plan_symbols(data, V).tolist()
    i = 0
    for t in ids:
        cands = [(v, k) for k, v in vocab.items() if len(k) >= 2 and data.startswith(k, i)]
        want = max(cands)[0] if cands else vocab[data[i:i + 1]]
        assert t == want, i
        i += len(next(k for k, v in vocab.items() if v == t))
    assert i == len(data)


def test_complete_fallback_to_literals(czo, wt, ot):
    """a byte sequence the vocabulary cannot spell (a lone byte >= 0x80: JSON keys are strings) makes encode_bytes fail, and the whole
    input becomes literal symbols vocab + byte (src/main.rs:863-868)"""
    for data in (b"abc\xffdef", b"\x80", b"caf\xc3", b"x" * 50 + b"\xfe\xff"):
        ids = wt.plan_symbols(data, V)
        assert ids.tolist() == [V + b for b in data] == czo.plan_rwkv_symbols(data, ot, V)
        assert wt.detok(ids, V) == data
    # a multi-byte character that IS in the vocabulary is a token, not a fallback
    e_acute = "é".encode()
    ids = wt.plan_symbols(b"caf" + e_acute, V)
    assert ids.max() < V and wt.detok(ids, V) == b"caf" + e_acute


def test_uncovered_tail_becomes_literals(cz, czo, tmp_path):
    """plan_rwkv_symbols re-measures the coverage with decode_bytes (src/main.rs:843-859).  With two keys sharing one id the decoded
    length can fall short of the input: the tail is then appended as literal symbols, and the result can be LONGER than the input.
    (The reference's choice between the two keys is its HashMap's; library and restatement both keep the last key in file order.)"""
    p = tmp_path / "dup.json"
    p.write_text('{"a": 1, "b": 2, "c": 3, "bc": 9, "": 1}')                 # id 1: the empty key survives in idx2token
    t = cz.WorldTokenizer(str(p)); o = czo.WorldTokenizer(str(p))
    data = b"abcab"
    want = czo.plan_rwkv_symbols(data, o, 100)
    assert want == [1, 9, 1, 2] + [100 + b for b in data[3:]]               # decoded length 0 + 2 + 0 + 1 = 3 < 5
    with pytest.raises(cz.CandlezipError):                                   # 6 symbols for 5 bytes: ids_cap is enforced
        t.plan_symbols(data, 100)
    assert t.plan_symbols(data, 100, ids_cap=6).tolist() == want
    assert t.detok(np.array(want, np.uint32), 100) == czo.rwkv_detok_with_literals(o, want, 100) == b"bcb" + data[3:]
    t.close()


def test_detok_mixes_tokens_and_literals(czo, wt, ot):
    syms = [V + 0xff, 5, 6, V + 0x00, V + 0x41, 300, 7, 99999 + V]           # unknown ids < vocab contribute nothing; literal = low byte
    a = np.array(syms, np.uint32)
    assert wt.detok(a, V) == czo.rwkv_detok_with_literals(ot, syms, V)
    assert wt.detok(np.zeros(0, np.uint32), V) == b""
    assert wt.plan_symbols(b"", V).size == 0


def test_open_errors(cz, tmp_path):
    with pytest.raises(cz.CandlezipError):
        cz.WorldTokenizer(str(tmp_path / "missing.json"))
    bad = tmp_path / "bad.json"; bad.write_text("[1, 2]")
    with pytest.raises(cz.CandlezipError):
        cz.WorldTokenizer(str(bad))

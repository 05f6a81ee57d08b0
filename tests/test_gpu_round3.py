"""Round-3 GPU parity tests: BASELINE config 5's schedule where it lives (context 1024 / reprime 1024: the KV sawtooth 511 -> 1535), the
context range of the reference (re-prime intervals whose streams pass 2048 positions, up to max_position_embeddings), the graph re-capture
branch of the sliced decode, and RWKV7's sequence-parallel encode."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def _check_streams(cz, g, o, streams, ctx, R, what, decode_ref=True):
    pairs = g.encode_tokens_batch(streams, ctx, R)
    pays = []
    for s_, (lo, hi) in zip(streams, pairs):
        ref_p, clo, chi, zw = o.encode_tokens(s_, ctx, R)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi), what
        p, zw2 = cz.ac_encode_pairs(lo, hi)
        assert p == ref_p and zw == zw2, what
        pays.append((p, zw))
    return pays


@pytest.mark.parametrize("form", ["narrow", "wide"])
def test_config5_schedule_reaches_1535_keys_both_step_forms(cz, czo, form):
    """--context 1024 --reprime-interval 1024 (BASELINE config 5; src/main.rs:1935, 2280-2284, 2506): the window is still 511, re-primes fall at
    i = 1024, 2048, so with 2 300 tokens the KV cache runs 511 -> 1535 keys twice.  Coder intervals and payload against the oracle; lock-step
    decode through the narrow step (dec_gemm_kernel, what B < 256 takes) and the wide one (gemm_mid_kernel, what the bench's 512-stream
    groups take), graph replay and eager."""
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=4)
    o = czo.SmolLmSession(czo.tiny_llama_config(), seed=11, embed_std=0.05)
    if form == "wide":
        g.debug_set("wide_min_streams", 1); g.debug_set("wide_mode", 3)
    rng = np.random.default_rng(505)
    lens = [2300, 1537, 2049]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pays = _check_streams(cz, g, o, streams, 1024, 1024, form)
    for no_graph in (0, 1):
        g.debug_set("no_graph", no_graph)
        outs = g.decode_tokens_batch([p for p, _ in pays], lens, 1024, 1024, 0)
        for s_, out, (p, zw) in zip(streams, outs, pays):
            if zw == 0:
                assert np.array_equal(out, s_), (form, no_graph)
    assert np.array_equal(outs[0], o.decode_tokens(pays[0][0], lens[0], 1024, 1024, 0))
    assert g.debug_get("cdf_fallback_rows") == 0
    g.close(); o.close()


def test_config5_schedule_135m_shape_properties(cz):
    """the same schedule at the SmolLM2-135M shape for 4 608 tokens (four re-primes, 1535 keys before each), where the oracle would need an
    hour: (1) window-by-window range encode (windows = the re-prime interval) + incremental coder == the one-shot payload; (2) a stream's
    intervals do not depend on its batch; (3) 3 + 2 streams on two lanes decode the input back."""
    T, W = 4608, 1024
    g = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=3)
    rng = np.random.default_rng(55)
    streams = [np.concatenate([[0], rng.integers(0, 49152, T)]).astype(np.uint32) for _ in range(5)]
    pairs = g.encode_tokens_batch(streams, 1024, 1024)
    payloads = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    encs = [cz.AcEncoder() for _ in streams]
    pieces = [b""] * len(streams)
    for w in range(-(-T // W)):
        part = g.encode_tokens_range_batch(streams, [(w * W, min((w + 1) * W, T))] * len(streams), 1024, 1024)
        for k, (lo, hi) in enumerate(part):
            pieces[k] += encs[k].push(lo, hi)
    for k in range(len(streams)):
        pieces[k] += encs[k].finish()[0]
        encs[k].close()
    assert [hashlib.sha256(p).hexdigest() for p in pieces] == [hashlib.sha256(p).hexdigest() for p in payloads]
    alone = g.encode_tokens_batch(streams[2:3], 1024, 1024)[0]
    assert np.array_equal(alone[0], pairs[2][0]) and np.array_equal(alone[1], pairs[2][1])
    g.debug_set("max_lanes", 2)
    outs = g.decode_tokens_batch(payloads, [T] * len(streams), 1024, 1024, 0)
    for s_, o_ in zip(streams, outs):
        assert np.array_equal(s_, o_)
    g.close()


@pytest.mark.parametrize("form", ["narrow", "wide"])
def test_reprime_interval_past_2048_positions(cz, czo, form):
    """the reference puts no bound on --reprime-interval below max_position_embeddings (src/main.rs:1935, 2280-2284): with interval 3000 a
    stream's KV cache reaches 511 + 3000 = 3511 positions.  Encode = one 3 511-row causal segment (attention_v3_kernel), decode = the
    long-context dec_attn variant on a KV cache sized for the run; all against the oracle."""
    gcfg, ocfg = cz.tiny_llama_config(max_pos=8192), czo.tiny_llama_config(max_pos=8192)
    g = cz.Session("smollm", gcfg, seed=11, embed_std=0.05, max_streams=3)
    o = czo.SmolLmSession(ocfg, seed=11, embed_std=0.05)
    if form == "wide":
        g.debug_set("wide_min_streams", 1); g.debug_set("wide_mode", 3)
    rng = np.random.default_rng(3000)
    lens = [3700, 2100, 600]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pays = _check_streams(cz, g, o, streams, 512, 3000, form)
    for no_graph in (0, 1):
        g.debug_set("no_graph", no_graph)
        outs = g.decode_tokens_batch([p for p, _ in pays], lens, 512, 3000, 0)
        for s_, out, (p, zw) in zip(streams, outs, pays):
            if zw == 0:
                assert np.array_equal(out, s_), (form, no_graph)
    assert np.array_equal(outs[0], o.decode_tokens(pays[0][0], lens[0], 512, 3000, 0))
    g.close(); o.close()


def test_trait_steps_past_the_initial_kv_cache(cz, czo):
    """the session trait (src/models.rs:92-119) across the cache's first 2048 positions: a 2 040-token re-prime, then single steps to position
    2 060 — the cache is re-laid out longer with its keys kept — every logit against the oracle; a position past max_position_embeddings is
    refused, as candle's RoPE table would."""
    gcfg, ocfg = cz.tiny_llama_config(max_pos=2304), czo.tiny_llama_config(max_pos=2304)
    g = cz.Session("smollm", gcfg, seed=11, embed_std=0.05, max_streams=2)
    o = czo.SmolLmSession(ocfg, seed=11, embed_std=0.05)
    rng = np.random.default_rng(9)
    toks = rng.integers(0, 512, 4200).astype(np.uint32)
    a = g.reprime_with_history_and_get_last_logits_tensor(toks[:2040]); b = o.reprime_logits(toks[:2040])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for i in range(2040, 2062):
        a = g.step_logits_tensor(toks[i]); b = o.step_logits(toks[i])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    assert g.index_pos() == o.index_pos() == 2062
    g.reprime_with_history_and_get_last_logits_tensor(toks[:2303]); g.step_logits_tensor(toks[5])     # positions 0..2303: the last one allowed
    with pytest.raises(cz.CandlezipError, match="max_position_embeddings"):
        g.step_logits_tensor(toks[6])
    with pytest.raises(cz.CandlezipError, match="max_position_embeddings"):
        g.encode_tokens(np.concatenate([[0], toks[:4100]]).astype(np.uint32), 512, 2000)               # the segment after i = 2000 would run to 511 + 2000 > 2304 positions
    with pytest.raises(cz.CandlezipError, match="max_position_embeddings"):
        g.decode_tokens(b"\x00" * 64, 4100, 512, 2000, 0)
    lo, hi = g.encode_tokens(np.concatenate([[0], toks[:2399]]).astype(np.uint32), 512, 2000)          # ... but a file that ends before that is fine, as in the reference
    assert lo.size == 2399
    g.close(); o.close()


def test_encode_between_decode_slices_recaptures_the_step_graph(cz):
    """an encode on the SAME session between two slices of a graph-replayed decode grows its work buffers: the captured step graph holds stale
    pointers and is re-captured (session.hip, decode_llama_advance); ids must equal the one-shot decode"""
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=3)
    rng = np.random.default_rng(77)
    lens = [900, 640]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    ref = g.decode_tokens_batch(pays, lens)
    g2 = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=3)     # fresh session: small buffers at first
    g2.decode_begin_batch(pays, lens)
    g2.decode_advance(300)
    big = [np.concatenate([[0], rng.integers(0, 512, 2500)]).astype(np.uint32) for _ in range(6)]
    other = g2.encode_tokens_batch(big)                                                            # grows x / xn / qkv / h ... under the captured graph
    g2.decode_advance(250)
    g2.encode_tokens_batch(big[:1])
    g2.decode_advance(10_000)
    outs = g2.decode_fetch_batch()
    g2.decode_end()
    for a, b, s_ in zip(outs, ref, streams):
        assert np.array_equal(a, b) and np.array_equal(a, s_)
    chk = g.encode_tokens_batch(big)
    for (lo, hi), (lo2, hi2) in zip(other, chk):
        assert np.array_equal(lo, lo2) and np.array_equal(hi, hi2)
    g.close(); g2.close()


@pytest.mark.parametrize("block_rows", [7, 64, 32768])
def test_rwkv7_sequence_parallel_encode_equals_oracle_and_lockstep_walk(cz, czo, block_rows):
    """RWKV7 encode, sequence-parallel (all tokens of a time block through the GEMMs at once, only the WKV scan sequential in t;
    candle_rwkv7/src/models/rwkv7.rs:179-328, 412-430; src/main.rs:2303-2325, 2345-2349) against the oracle's token-by-token walk and against
    the library's own lock-step walk: coder intervals and payload bytes.  Literals (coded, the model is not stepped), ragged and degenerate
    streams, more streams than one group, time blocks of 7 / 64 rows (the token shift and the state cross block boundaries) and one block."""
    g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=4)
    o = czo.Rwkv7Session(czo.tiny_rwkv7_config(), seed=9, embed_std=0.05)
    rng = np.random.default_rng(404)
    lens = [150, 1, 0, 64, 97, 33, 2]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    streams[0][3] = 512 + 10; streams[0][4] = 512 + 255; streams[0][149] = 512; streams[4][1] = 512 + 7; streams[5][33] = 512 + 99   # literal escapes, first / consecutive / last
    g.debug_set("rwkv_seq", 1); g.debug_set("rwkv_seq_rows", block_rows)
    seq = g.encode_tokens_batch(streams)
    g.debug_set("rwkv_seq", 0)
    walk = g.encode_tokens_batch(streams)
    g.debug_set("rwkv_seq", 1); g.debug_set("rwkv_seq_rows", 32768)
    pays = []
    for s_, (lo, hi), (lo2, hi2) in zip(streams, seq, walk):
        ref_p, clo, chi, zw = o.encode_tokens(s_)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi), (block_rows, s_.size)
        assert np.array_equal(lo2, clo) and np.array_equal(hi2, chi)
        p, zw2 = cz.ac_encode_pairs(lo, hi)
        assert p == ref_p and zw == zw2
        pays.append(p)
    outs = g.decode_tokens_batch(pays, lens)
    for s_, out in zip(streams, outs):
        assert np.array_equal(out, s_)
    assert g.debug_get("cdf_fallback_rows") == 0
    g.close(); o.close()


def test_rwkv7_0b1_shape_sequence_parallel_encode(cz, czo):
    """the RWKV7-0.1B shape: 24 tokens of two streams against the oracle (every GEMM shape of the sequence-parallel path at its real K / N),
    then 3 000 tokens x 3 streams: sequence-parallel == lock-step walk, and the round trip"""
    g = cz.Session("rwkv7", cz.rwkv7_0b1_config(), seed=3, embed_std=0.02, max_streams=3)
    o = czo.Rwkv7Session(czo.rwkv7_0b1_config(), seed=3, embed_std=0.02)
    rng = np.random.default_rng(12)
    small = [np.concatenate([[0], rng.integers(0, 65536, n)]).astype(np.uint32) for n in (14, 10)]
    small[0][5] = 65536 + 65
    for s_, (lo, hi) in zip(small, g.encode_tokens_batch(small)):
        ref_p, clo, chi, _ = o.encode_tokens(s_)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    o.close()
    big = [np.concatenate([[0], rng.integers(0, 65536 + 256, n)]).astype(np.uint32) for n in (3000, 2500, 700)]
    for b in big:
        b[0] = 0
    seq = g.encode_tokens_batch(big)
    g.debug_set("rwkv_seq", 0)
    walk = g.encode_tokens_batch(big)
    g.debug_set("rwkv_seq", 1)
    for (lo, hi), (lo2, hi2) in zip(seq, walk):
        assert np.array_equal(lo, lo2) and np.array_equal(hi, hi2)
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in seq]
    outs = g.decode_tokens_batch(pays, [b.size - 1 for b in big])
    for b, out in zip(big, outs):
        assert np.array_equal(out, b)
    g.close()


@pytest.mark.parametrize("shape", ["tiny", "0b1"])
def test_rwkv7_wide_group_step_is_bitexact(cz, czo, shape):
    """the RWKV7 lock-step step of a WIDE group (natural-layout activations, every projection on the LDS-tiled MFMA GEMM, the lm_head masked
    for streams whose symbol was a literal) against the oracle: trait-step logits, the lock-step encode walk, decode with literals, graph and eager"""
    if shape == "tiny":
        gcfg, ocfg, V, n_tok, seed, std = cz.tiny_rwkv7_config(), czo.tiny_rwkv7_config(), 512, 130, 9, 0.05
    else:
        gcfg, ocfg, V, n_tok, seed, std = cz.rwkv7_0b1_config(), czo.rwkv7_0b1_config(), 65536, 10, 3, 0.02
    g = cz.Session("rwkv7", gcfg, seed=seed, embed_std=std, max_streams=5)
    o = czo.Rwkv7Session(ocfg, seed=seed, embed_std=std)
    g.debug_set("rwkv_wide_min", 1); g.debug_set("rwkv_seq", 0)
    rng = np.random.default_rng(8)
    toks = rng.integers(0, V, 12).astype(np.uint32)
    o.reset()
    a = g.reprime_with_history_and_get_last_logits_tensor(toks[:1]); b = o.step_logits(toks[0])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for t in toks[1:6]:
        a = g.step_logits_tensor(t); b = o.step_logits(t)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    lens = [n_tok, n_tok // 2, 1, 0, n_tok - 3]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    streams[0][2] = V + 65; streams[0][3] = V + 1; streams[4][n_tok - 4] = V + 200
    pays = []
    for s_, (lo, hi) in zip(streams, g.encode_tokens_batch(streams)):
        ref_p, clo, chi, zw = o.encode_tokens(s_)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
        pays.append(ref_p)
    for no_graph in (0, 1):
        g.debug_set("no_graph", no_graph)
        for s_, out in zip(streams, g.decode_tokens_batch(pays, lens)):
            assert np.array_equal(out, s_), no_graph
    g.close(); o.close()


@pytest.mark.parametrize("shape", ["tiny", "135m"])
def test_split_decode_attention_of_small_groups_is_bitexact(cz, czo, shape):
    """the three-launch decode attention small groups take (scores per (head, chunk) block; softmax statistics + P.V partials; ordered sum) against
    the oracle and against the one-kernel forms: trait steps up to a context of 130 keys (three chunks), a lock-step decode of ragged streams across a
    re-prime (511 -> 700 keys), graph replay and eager"""
    if shape == "tiny":
        gcfg, ocfg, V, n_tok, seed, std = cz.tiny_llama_config(), czo.tiny_llama_config(), 512, 700, 11, 0.05
    else:
        gcfg, ocfg, V, n_tok, seed, std = cz.smollm2_135m_config(), czo.smollm2_135m_config(), 49152, 70, 0x5EED, 0.02
    g = cz.Session("smollm", gcfg, seed=seed, embed_std=std, max_streams=3)
    o = czo.SmolLmSession(ocfg, seed=seed, embed_std=std)
    rng = np.random.default_rng(33)
    toks = rng.integers(0, V, 140).astype(np.uint32)
    g.debug_set("dec_attn_split_max", 128)
    a = g.reprime_with_history_and_get_last_logits_tensor(toks[:60]); b = o.reprime_logits(toks[:60])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for t in toks[60:132]:                                                       # 61 .. 132 keys: the chunk count goes 1 -> 3 on the way
        a = g.step_logits_tensor(t); b = o.step_logits(t)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    lens = [n_tok, n_tok // 3, 2]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    coded = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
    ref = o.decode_tokens(coded[0][0], lens[0])
    outs = {}
    for split in (128, 0):
        g.debug_set("dec_attn_split_max", split)
        for no_graph in (0, 1):
            g.debug_set("no_graph", no_graph)
            outs[(split, no_graph)] = g.decode_tokens_batch([p for p, _ in coded], lens)
            assert np.array_equal(outs[(split, no_graph)][0], ref), (split, no_graph)
    for k in outs:
        for x, y, (_, z), s_ in zip(outs[k], outs[(0, 1)], coded, streams):
            assert np.array_equal(x, y)
            if z == 0:
                assert np.array_equal(x, s_)
    g.close(); o.close()                                                         # (dec_attn_split_max: reset by conftest's fixture finalizer)

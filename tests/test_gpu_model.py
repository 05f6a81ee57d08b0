"""GPU parity tests for the whole hot path through the C ABI: session trait, segment-parallel encode, lock-step decode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def _eq(a, b, what):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32); b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    n = int(np.sum(a != b))
    assert n == 0, f"{what}: {n}/{a.size} logits differ"


@pytest.fixture(scope="module")
def tiny_pair(cz, czo):
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=8)
    o = czo.SmolLmSession(czo.tiny_llama_config(), seed=11, embed_std=0.05)
    yield g, o
    g.close(); o.close()


def test_session_trait_step_and_reprime_bitexact(cz, czo, tiny_pair):
    g, o = tiny_pair
    assert g.vocab_size() == o.vocab_size() == 512 and g.max_context_length() == 512
    rng = np.random.default_rng(0)
    toks = rng.integers(0, 512, 90).astype(np.uint32)
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.reprime_logits(toks[:1]), "reprime(1)")
    for i, t in enumerate(toks[1:70]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"step {i}")
        assert g.index_pos() == o.index_pos() == i + 2
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[10:87]), o.reprime_logits(toks[10:87]), "reprime(77)")
    assert g.index_pos() == 77
    _eq(g.step_logits_tensor(toks[87]), o.step_logits(toks[87]), "step after reprime")


def test_error_paths_mirror_reference(cz, tiny_pair):
    g, _ = tiny_pair
    with pytest.raises(cz.CandlezipError, match="empty history"):
        g.reprime_with_history_and_get_last_logits_tensor(np.zeros(0, np.uint32))
    with pytest.raises(cz.CandlezipError, match="vocab"):
        g.step_logits_tensor(512)
    with pytest.raises(cz.CandlezipError):
        cz.Session("smollm", cz.LlamaConfig(512, 100, 2, 2, 1, 256, 2048, 1e5, 1e-5))      # head_dim != 64


def test_encode_pairs_and_payload_identical_to_oracle(cz, czo, tiny_pair):
    g, o = tiny_pair
    rng = np.random.default_rng(1)
    ids = np.concatenate([[0], rng.integers(0, 512, 1300)]).astype(np.uint32)      # crosses two re-prime boundaries
    payload, clo, chi, zw = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    pay, zw2 = cz.ac_encode_pairs(lo, hi)
    assert pay == payload and zw2 == zw == 0                                           # bitstream-identical to the CPU path
    out = g.decode_tokens(payload, 1300)
    assert np.array_equal(out, ids)


def test_real_text_ids_canz_identical_to_oracle(cz, czo, golden):
    """BASELINE config 2 in miniature: the first 1200 token ids of the reference's own asyoulik.txt decode trace (tests/golden,
    SmolLM2 vocabulary 49 152) through a small seeded model: the GPU `.canz` (header v2 + payload) is byte-identical to the CPU
    oracle's, and the GPU decodes it back to the ids."""
    import os
    ids = np.fromfile(os.path.join(golden, "tokens_asyoulik.u16"), np.uint16).astype(np.uint32)
    assert ids.size == 39915 and ids.max() < 49152
    x = np.concatenate([[0], ids[:1200]]).astype(np.uint32)
    cfg_g = cz.tiny_llama_config(vocab=49152, d_model=128, n_layers=2, n_heads=2, n_kv_heads=1, d_ffn=256)
    cfg_o = czo.tiny_llama_config(vocab=49152, d_model=128, n_layers=2, n_heads=2, n_kv_heads=1, d_ffn=256)
    g = cz.Session("smollm", cfg_g, seed=5, embed_std=0.05, max_streams=2)
    o = czo.SmolLmSession(cfg_o, seed=5, embed_std=0.05)
    payload, clo, chi, zw = o.encode_tokens(x)
    lo, hi = g.encode_tokens(x)
    pay, zw2 = cz.ac_encode_pairs(lo, hi)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi) and pay == payload and zw == zw2 == 0
    text = open(os.path.join(golden, "inputs", "asyoulik.txt"), "rb").read()
    blobs = []
    for mod, H in ((cz, cz.Header), (czo, czo.Header)):
        h = H()
        h.bos_token_id, h.token_count, h.orig_len_bytes = 0, 1200, len(text)
        digest = mod.blake3_16(text)
        for dst in (h.model_hash16, h.tokenizer_hash16, h.orig_hash16):
            for i in range(16):
                dst[i] = digest[i]
        h.reserved_flags, h.context_window, h.vocab_size, h.reprime_interval = 512 << 16, 512, 49152, 512
        blobs.append(mod.write_header(h, b"synthetic:5") + (pay if mod is cz else payload))
    assert blobs[0] == blobs[1] and blobs[0][:4] == b"GTPZ"
    hh, rep, used = cz.read_header(blobs[0])
    assert rep == b"synthetic:5" and hh.token_count == 1200
    out = g.decode_tokens(blobs[0][used:], 1200)
    assert np.array_equal(out, x)
    # the span scorer on the same real ids at the full vocabulary width (f4)
    assert g.xe_bits_over_span(x[:700], x[700:760], x[900:930]) == o.xe_bits_over_span(x[:700], x[700:760], x[900:930])[0]
    g.close(); o.close()


def test_cross_entropy_bits_over_span_matches_oracle(cz, czo, tiny_pair):
    """cross_entropy_bits_over_span (src/main.rs:1725-1751).  The probabilities are reproduced bit for bit (floor-softmax
    chains evaluated exactly in parallel); -log2 and the sum are libm on both sides, so the bits are compared for equality too."""
    g, o = tiny_pair
    rng = np.random.default_rng(17)
    hist = rng.integers(0, 512, 900).astype(np.uint32)
    cases = [(hist, rng.integers(0, 512, 64), ()), (hist[:40], rng.integers(0, 512, 5), ()), (hist, rng.integers(0, 512, 128), rng.integers(0, 512, 48)),
             ((), rng.integers(0, 512, 9), rng.integers(0, 512, 600)),                    # hint longer than the 511 budget: truncated, no history left
             (hist, (), ()), (hist[:1], [7], ()), (hist, rng.integers(0, 512, 700), rng.integers(0, 512, 100))]
    want = [o.xe_bits_over_span(h, t, hint)[0] for h, t, hint in cases]
    got = g.xe_bits_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    assert got.tolist() == want
    assert g.xe_bits_over_span(hist, cases[0][1]) == want[0] and want[4] == 0.0
    with pytest.raises(cz.CandlezipError) as e:
        g.xe_bits_over_span((), [1, 2, 3])
    assert "empty history" in str(e.value)                                              # the reference's reprime error
    # a target outside the vocabulary scores ac_p_min (pdf.get(sym).unwrap_or(p_min)) in the oracle; the device path refuses ids it cannot feed
    assert g.debug_get("cdf_fallback_rows") == 0


def test_session_fork_shares_weights(cz, czo, tiny_pair):
    """cz_session_fork: a second independent session on the same device tensors — same bits, independent state, usable from another
    host thread while the parent works (the reference loads the model twice for its second session, src/main.rs:1943-1956)"""
    import threading
    g, o = tiny_pair
    rng = np.random.default_rng(23)
    a = np.concatenate([[0], rng.integers(0, 512, 900)]).astype(np.uint32); b = np.concatenate([[0], rng.integers(0, 512, 700)]).astype(np.uint32)
    f = g.fork()
    assert f.vocab_size() == g.vocab_size()
    ref_a = g.encode_tokens(a); pay_a = cz.ac_encode_pairs(*ref_a)[0]
    out = {}
    t = threading.Thread(target=lambda: out.setdefault("b", f.encode_tokens(b)))           # the fork compresses ...
    t.start()
    dec_a = g.decode_tokens(pay_a, a.size - 1)                                              # ... while the parent decompresses
    t.join()
    assert np.array_equal(dec_a, a)
    ref_b = g.encode_tokens(b)
    assert np.array_equal(out["b"][0], ref_b[0]) and np.array_equal(out["b"][1], ref_b[1])
    _eq(f.reprime_with_history_and_get_last_logits_tensor(a[:50]), g.reprime_with_history_and_get_last_logits_tensor(a[:50]), "fork reprime")
    assert f.index_pos() == 50
    g.step_logits_tensor(5)
    assert g.index_pos() == 51 and f.index_pos() == 50                                      # independent state
    with pytest.raises(cz.CandlezipError):
        f.fork()
    f.close()
    # a fork that outlives its parent is a husk: every call fails loudly, closing it is still safe
    p2 = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=2)
    f2 = p2.fork()
    p2.close()
    with pytest.raises(cz.CandlezipError) as e:
        f2.encode_tokens(a)
    assert "parent" in str(e.value)
    f2.close()


def test_context_1024_and_small_intervals(cz, czo, tiny_pair):
    g, o = tiny_pair
    rng = np.random.default_rng(2)
    ids = np.concatenate([[0], rng.integers(0, 512, 1100)]).astype(np.uint32)
    for ctx, R in ((1024, 1024), (512, 256), (512, 100)):
        p, clo, chi, _ = o.encode_tokens(ids, context=ctx, reprime_interval=R)
        lo, hi = g.encode_tokens(ids, context=ctx, reprime_interval=R)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi), (ctx, R)
        assert np.array_equal(g.decode_tokens(p, 1100, context_window=ctx, reprime_interval=R), ids), (ctx, R)


def test_batched_ragged_streams_roundtrip(cz, czo, tiny_pair):
    g, o = tiny_pair
    rng = np.random.default_rng(3)
    lens = [1, 2, 50, 513, 700, 1025]                      # incl. BOS-only and exactly-at-boundary streams
    streams = [np.concatenate([[0], rng.integers(0, 512, n - 1)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    payloads = []
    for s, (lo, hi) in zip(streams, pairs):
        ref_p, clo, chi, _ = o.encode_tokens(s)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
        p, _ = cz.ac_encode_pairs(lo, hi)
        assert p == ref_p
        payloads.append(p)
    outs = g.decode_tokens_batch(payloads, [n - 1 for n in lens])
    for s, out in zip(streams, outs):
        assert np.array_equal(out, s)


def test_zero_width_hazard_is_reproduced_and_counted(cz, czo):
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=5, embed_std=1.5, max_streams=2)
    o = czo.SmolLmSession(czo.tiny_llama_config(), seed=5, embed_std=1.5)
    rng = np.random.default_rng(4)
    ids = np.concatenate([[0], rng.integers(0, 512, 200)]).astype(np.uint32)
    payload, clo, chi, zw = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    pay, zw2 = cz.ac_encode_pairs(lo, hi)
    assert pay == payload and zw2 == zw > 0
    # decoding the undecodable stream must go wrong in exactly the way the reference decoder does
    assert np.array_equal(g.decode_tokens(payload, 200), o.decode_tokens(payload, 200))
    g.close(); o.close()


def test_smollm2_135m_shape_bitexact(cz, czo):
    """the BASELINE model shape (SmolLM2-135M: d 576, 30 layers, 9/3 heads, ffn 1536, vocab 49152), synthetic weights"""
    g = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=4)
    o = czo.SmolLmSession(czo.smollm2_135m_config(), seed=0x5EED, embed_std=0.02)
    rng = np.random.default_rng(6)
    toks = rng.integers(0, 49152, 200).astype(np.uint32)
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.reprime_logits(toks[:1]), "bos step")
    for i, t in enumerate(toks[1:6]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"step {i}")
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:150]), o.reprime_logits(toks[:150]), "reprime(150)")
    _eq(g.step_logits_tensor(toks[150]), o.step_logits(toks[150]), "step@150")
    ids = np.concatenate([[0], toks[:60]]).astype(np.uint32)
    payload, clo, chi, zw = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    pay, _ = cz.ac_encode_pairs(lo, hi)
    assert pay == payload
    outs = g.decode_tokens_batch([payload, payload[:]], [60, 60])
    assert np.array_equal(outs[0], ids) and np.array_equal(outs[1], ids)
    g.close(); o.close()


def test_other_llama_family_shape_bitexact(cz, czo):
    """a second LLaMA-family shape through the same kernels (SmolLM2-360M's geometry: d 960, 15/5 heads, ffn 2560, rope 1e5; 6 of its 32
    layers to keep the CPU oracle quick): different K-chunk counts (15 and 40), odd tile counts, GQA group of 3, V = 49 152"""
    cfg_g = cz.LlamaConfig(49152, 960, 6, 15, 5, 2560, 8192, 100000.0, 1e-5)
    cfg_o = czo.LlamaConfig(49152, 960, 6, 15, 5, 2560, 8192, 100000.0, 1e-5)
    g = cz.Session("smollm", cfg_g, seed=77, embed_std=0.02, max_streams=3)
    o = czo.SmolLmSession(cfg_o, seed=77, embed_std=0.02)
    rng = np.random.default_rng(8)
    toks = rng.integers(0, 49152, 700).astype(np.uint32)
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.reprime_logits(toks[:1]), "bos step")
    for i, t in enumerate(toks[1:4]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"step {i}")
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:130]), o.reprime_logits(toks[:130]), "reprime(130)")
    ids = np.concatenate([[0], toks[:40]]).astype(np.uint32)
    payload, clo, chi, zw = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi) and cz.ac_encode_pairs(lo, hi)[0] == payload
    # three ragged streams in lock-step, one of them long enough to re-prime (GPU round trip; the oracle already pinned the pairs above)
    streams = [np.concatenate([[0], toks[:n]]).astype(np.uint32) for n in (40, 600, 7)]
    pays = [cz.ac_encode_pairs(*g.encode_tokens(s_))[0] for s_ in streams]
    outs = g.decode_tokens_batch(pays, [s_.size - 1 for s_ in streams])
    for s_, out in zip(streams, outs):
        assert np.array_equal(s_, out)
    g.close(); o.close()


def test_decode_paths_agree_graph_eager_v1(cz, czo, tiny_pair):
    """hipGraph replay, eager fast kernels and the general (v1) kernels must decode identical ids"""
    g, o = tiny_pair
    rng = np.random.default_rng(8)
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in (700, 300, 1200)]
    pays = []
    for s in streams:
        lo, hi = g.encode_tokens(s)
        pays.append(cz.ac_encode_pairs(lo, hi)[0])
    for knobs in ({}, {"no_graph": 1}, {"decode_v1": 1}, {"serial_cdf": 1}, {"attn_v1": 1}):
        for k, v in knobs.items():
            g.debug_set(k, v)
        outs = g.decode_tokens_batch(pays, [s.size - 1 for s in streams])
        for k in knobs:
            g.debug_set(k, 0)
        for s, out in zip(streams, outs):
            assert np.array_equal(s, out), knobs
    assert g.debug_get("cdf_fallback_rows") == 0


def test_decode_lanes_and_fresh_session(cz, czo, tiny_pair):
    """more streams than one lock-step group holds: the groups run concurrently on decode lanes (own HIP stream, KV cache and
    step graph, shared weights) and must return exactly what one group at a time returns; the decoding session is FRESH (never
    encoded), so the buffers of its first re-prime are the ones reserved before the step graph is captured."""
    g, o = tiny_pair
    rng = np.random.default_rng(21)
    lens = [700, 40, 1300, 1, 0, 650, 1100, 513, 900, 64, 1250]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pays = []
    for s in streams:
        lo, hi = g.encode_tokens(s)
        pays.append(cz.ac_encode_pairs(lo, hi)[0])
    cfg = cz.tiny_llama_config()
    for lanes in (1, 2, 4):
        f = cz.Session("smollm", cfg, seed=11, embed_std=0.05, max_streams=3)     # 11 streams -> 4 groups of 3,3,3,2
        f.debug_set("max_lanes", lanes)
        outs = f.decode_tokens_batch(pays, [s.size - 1 for s in streams])
        outs2 = f.decode_tokens_batch(pays[::-1], [s.size - 1 for s in streams[::-1]])   # graphs and lanes are reused
        assert f.debug_get("cdf_fallback_rows") == 0
        f.close()
        for s, out, out2 in zip(streams, outs, outs2[::-1]):
            assert np.array_equal(s, out) and np.array_equal(s, out2), lanes


# ---------------------------------------------------------------- RWKV7 (config 4: recurrent-state kernel path)
@pytest.fixture(scope="module")
def rwkv_pair(cz, czo):
    g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=8)
    o = czo.Rwkv7Session(czo.tiny_rwkv7_config(), seed=9, embed_std=0.05)
    yield g, o
    g.close(); o.close()


def test_rwkv7_step_logits_bitexact(cz, czo, rwkv_pair):
    g, o = rwkv_pair
    assert g.vocab_size() == 512 and g.max_context_length() > 1 << 60          # usize::MAX (src/models.rs:149)
    rng = np.random.default_rng(0)
    toks = rng.integers(0, 512, 40).astype(np.uint32)
    o.reset()
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.step_logits(toks[0]), "rwkv step 0")
    for i, t in enumerate(toks[1:]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"rwkv step {i + 1}")
    # reprime = reset + sequential feed (src/models.rs:161-170)
    o.reset()
    ref = [o.step_logits(t) for t in toks[:9]][-1]
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:9]), ref, "rwkv reprime(9)")


def test_rwkv7_encode_decode_bitexact_with_literals(cz, czo, rwkv_pair):
    g, o = rwkv_pair
    rng = np.random.default_rng(1)
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in (120, 40, 75)]
    streams[1][5] = 512 + 65; streams[1][9] = 512 + 255                        # literal escapes: coded, but do not step the model
    pairs = g.encode_tokens_batch(streams)
    pays = []
    for s, (lo, hi) in zip(streams, pairs):
        ref_p, clo, chi, zw = o.encode_tokens(s)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
        p, zw2 = cz.ac_encode_pairs(lo, hi)
        assert p == ref_p and zw2 == zw == 0
        pays.append(p)
    outs = g.decode_tokens_batch(pays, [s.size - 1 for s in streams])
    for s, out in zip(streams, outs):
        assert np.array_equal(out, s)


def test_rwkv7_cross_entropy_bits_over_span_matches_oracle(cz, czo, rwkv_pair):
    """cross_entropy_bits_over_span_rwkv (src/main.rs:1753-1787): the history loses its literal symbols, nothing is cut (max_context_length() is
    usize::MAX), p = combined_pdf_with_literals(logits)[target] (three exact chains on the device), ac_p_min() for a target past vocab + 256, the
    model steps on vocabulary targets only.  Same probabilities bit for bit, so (libm log2 and the running sum on the host on both sides) same bits."""
    g, o = rwkv_pair
    rng = np.random.default_rng(23)
    V = 512
    hist = rng.integers(0, V, 300).astype(np.uint32)
    hist_lit = hist.copy(); hist_lit[[3, 50, 51, 299]] = [V + 7, V + 255, V + 0, V + 100]          # literals in the history: filtered out
    t_mix = rng.integers(0, V, 40).astype(np.uint32); t_mix[[0, 5, 6, 39]] = [V + 1, V + 200, V + 255, V + 9]   # literal targets: scored, no step
    t_far = rng.integers(0, V, 12).astype(np.uint32); t_far[[2, 11]] = [V + 256, V + 5000]           # past the pdf's end: ac_p_min(), no step
    cases = [(hist, rng.integers(0, V, 64), ()), (hist_lit, t_mix, ()), (hist[:1], [7], ()), (hist, t_far, rng.integers(0, V, 20)),
             ((), rng.integers(0, V, 9), rng.integers(0, V, 33)), (hist, (), ()), (hist_lit[:60], [V + 3, V + 4], [5]),
             (hist, rng.integers(0, V, 700), ()), ([V + 1, 4, V + 2], [1, 2, 3], ())]
    want = [o.xe_bits_over_span(h, t, hint)[0] for h, t, hint in cases]
    got = g.xe_bits_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
    assert got.tolist() == want
    assert want[5] == 0.0 and g.xe_bits_over_span(hist_lit, t_mix) == want[1]
    for h, t, hint, msg in (((), [1, 2, 3], (), "empty history"), ([V + 3, V + 9], [1], (), "empty history"), (hist, [1], [V], "hint token")):
        with pytest.raises(cz.CandlezipError) as e:
            g.xe_bits_over_span(h, t, hint)
        assert msg in str(e.value)
    assert g.debug_get("cdf_fallback_rows") == 0
    # the stepping trait is still usable afterwards (the scorer resets the recurrent state, like the reference's reprime does)
    o.reset()
    _eq(g.reprime_with_history_and_get_last_logits_tensor(hist[:5]), [o.step_logits(t) for t in hist[:5]][-1], "rwkv reprime after xe")


def test_rwkv7_lanes_match_single_group(cz, czo, rwkv_pair):
    """more streams than one lock-step group: RWKV encode AND decode run the groups concurrently on lanes (shared weights, own
    recurrent state); pairs, payloads and ids must equal the oracle's stream by stream, whatever the lane count"""
    g0, o = rwkv_pair
    rng = np.random.default_rng(5)
    lens = [90, 31, 64, 1, 77, 50, 0, 12, 83]
    streams = [np.concatenate([[0], rng.integers(0, 512 + 256, n)]).astype(np.uint32) for n in lens]     # literals included
    refs = [o.encode_tokens(s) for s in streams]
    for lanes, no_graph in ((1, 0), (3, 0), (2, 1)):                                                       # hipGraph replay vs eager launches
        g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=2)             # 9 streams -> 5 groups
        g.debug_set("max_lanes", lanes)
        g.debug_set("no_graph", no_graph)
        pairs = g.encode_tokens_batch(streams)
        pays = []
        for (lo, hi), (ref_p, clo, chi, zw) in zip(pairs, refs):
            assert np.array_equal(lo, clo) and np.array_equal(hi, chi), lanes
            pays.append(cz.ac_encode_pairs(lo, hi)[0])
            assert pays[-1] == ref_p
        outs = g.decode_tokens_batch(pays, [s.size - 1 for s in streams])
        for s_, out in zip(streams, outs):
            assert np.array_equal(out, s_), lanes
        assert g.debug_get("cdf_fallback_rows") == 0
        g.close()


def test_rwkv7_0b1_shape_bitexact(cz, czo):
    """RWKV7-0.1B shape (C 768, 12 layers, 12 heads x 64, ffn 3072, vocab 65536), synthetic weights"""
    g = cz.Session("rwkv7", cz.rwkv7_0b1_config(), seed=3, embed_std=0.02, max_streams=2)
    o = czo.Rwkv7Session(czo.rwkv7_0b1_config(), seed=3, embed_std=0.02)
    rng = np.random.default_rng(2)
    toks = rng.integers(0, 65536, 6).astype(np.uint32)
    o.reset()
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.step_logits(toks[0]), "0.1B step 0")
    for i, t in enumerate(toks[1:]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"0.1B step {i + 1}")
    ids = np.concatenate([[0], toks[:4]]).astype(np.uint32)
    ref_p, clo, chi, _ = o.encode_tokens(ids)
    lo, hi = g.encode_tokens(ids)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    assert np.array_equal(g.decode_tokens(ref_p, 4), ids)
    # the span scorer at this vocabulary (V + 256 = 65 792 symbols: 68 per thread in the literals-mode kernel), a literal and an out-of-range target among them
    tg = np.array([toks[4], 65536 + 10, toks[5], 65536 + 256], np.uint32)
    assert g.xe_bits_over_span(toks[:4], tg) == o.xe_bits_over_span(toks[:4], tg)[0]
    g.close(); o.close()

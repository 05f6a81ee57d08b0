"""Round-2 GPU parity tests through the C ABI:
 * range (segment-sharded) encode == the matching slice of the whole-stream encode, for arbitrary ranges;
 * the decode loop in slices (begin / advance / fetch / end) == the one-shot decode, for both back-ends, graph replay and eager, several lanes;
 * the SmolLM2-135M shape on >= 1100 REAL token ids (the reference's own SmolLM2 tokenisation of asyoulik.txt, tests/golden/tokens_asyoulik.u16)
   across two re-prime boundaries: pairs + payload vs the oracle, then a three-lane decode;
 * BASELINE config 4: Canterbury files through the RWKV7 recurrent-state path (World tokenizer fixture), pairs vs the oracle on a prefix at
   the 0.1B shape and a byte-exact round trip of the whole files."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def test_range_encode_equals_slices_of_the_full_encode(cz):
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=4)
    rng = np.random.default_rng(3)
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in (1700, 600, 40)]
    full = g.encode_tokens_batch(streams, 512, 512)
    for ranges in ([(0, 1700), (0, 600), (0, 40)], [(512, 1024), (100, 101), (0, 0)], [(1024, 1700), (511, 600), (39, 40)], [(700, 1300), (0, 513), (5, 33)]):
        part = g.encode_tokens_range_batch(streams, ranges, 512, 512)
        for (b, e), (lo, hi), (flo, fhi) in zip(ranges, part, full):
            assert lo.size == e - b and np.array_equal(lo, flo[b:e]) and np.array_equal(hi, fhi[b:e]), (b, e)
    # another schedule (reprime every 256 once the window is full) and the same stream given twice with different ranges
    full = g.encode_tokens_batch(streams[:1], 300, 256)[0]
    a, b_ = g.encode_tokens_range_batch([streams[0], streams[0]], [(0, 768), (768, 1700)], 300, 256)
    assert np.array_equal(np.concatenate([a[0], b_[0]]), full[0]) and np.array_equal(np.concatenate([a[1], b_[1]]), full[1])
    with pytest.raises(cz.CandlezipError):
        g.encode_tokens_range_batch(streams[:1], [(10, 1701)], 512, 512)
    with pytest.raises(cz.CandlezipError, match="empty history"):
        g.encode_tokens(streams[0], 0, 512)                                        # context 0: the reference fails at its first re-prime
    g.close()
    r = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=2)
    with pytest.raises(cz.CandlezipError, match="no re-prime"):
        r.encode_tokens_range_batch(streams[:1], [(0, 10)])
    r.close()


@pytest.mark.parametrize("no_graph", [0, 1])
def test_sliced_decode_equals_one_shot_smollm(cz, no_graph):
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=3)
    g.debug_set("max_lanes", 3); g.debug_set("no_graph", no_graph)
    rng = np.random.default_rng(8)
    lens = [1400, 700, 1, 1203, 0, 512, 513, 90]                                   # 8 streams -> 3 lock-step groups, one per lane
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in g.encode_tokens_batch(streams)]
    ref = g.decode_tokens_batch(pays, lens)
    for s_, o in zip(streams, ref):
        assert np.array_equal(s_, o)
    g.decode_begin_batch(pays, lens)
    seen = 0
    for step in (1, 510, 1, 1, 300, 10 ** 9):                                      # slices that straddle the re-prime at i = 512 and the one at 1024
        pos, rem = g.decode_advance(step)
        seen = min(seen + step, max(lens))
        assert pos == seen and rem == max(lens) - seen
        part = g.decode_fetch_batch()
        for s_, o, n in zip(streams, part, lens):
            k = min(pos, n)
            assert np.array_equal(o[: k + 1], s_[: k + 1])
    g.decode_end()
    with pytest.raises(cz.CandlezipError, match="no decode in progress"):
        g.decode_advance(1)
    # more groups than lanes cannot stay resident
    g.debug_set("max_lanes", 2)
    with pytest.raises(cz.CandlezipError, match="resident"):
        g.decode_begin_batch(pays, lens)
    assert all(np.array_equal(a, b) for a, b in zip(g.decode_tokens_batch(pays, lens), ref))   # the one-shot path still cycles groups over lanes
    g.close()


def test_sliced_decode_equals_one_shot_rwkv7(cz):
    g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=2)
    g.debug_set("max_lanes", 2)
    rng = np.random.default_rng(4)
    lens = [90, 31, 64]
    streams = [np.concatenate([[0], rng.integers(0, 512 + 256, n)]).astype(np.uint32) for n in lens]
    pays = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in g.encode_tokens_batch(streams)]
    g.decode_begin_batch(pays, lens)
    for step in (7, 50, 1000):
        g.decode_advance(step)
    outs = g.decode_fetch_batch()
    g.decode_end()
    for s_, o in zip(streams, outs):
        assert np.array_equal(s_, o)
    g.close()


def test_smollm2_135m_shape_real_ids_across_two_reprimes(cz, czo, golden):
    """>= 1100 ids of the reference's own asyoulik.txt tokenisation at the BASELINE shape: the re-primes at i = 512 and i = 1024 are crossed
    against the oracle (pairs and payload bytes), then the payloads are decoded on three lanes"""
    ids = np.concatenate([[0], np.fromfile(os.path.join(golden, "tokens_asyoulik.u16"), "<u2")[:1150]]).astype(np.uint32)
    g = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=2)
    o = czo.SmolLmSession(czo.smollm2_135m_config(), seed=0x5EED, embed_std=0.02)
    payload, clo, chi, zw = o.encode_tokens(ids, 512, 512)
    lo, hi = g.encode_tokens(ids, 512, 512)
    assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    pay, zw2 = cz.ac_encode_pairs(lo, hi)
    assert pay == payload and zw2 == zw
    g.debug_set("max_lanes", 3)
    toks = np.fromfile(os.path.join(golden, "tokens_asyoulik.u16"), "<u2")
    streams = [ids, np.concatenate([[0], toks[2000:2700]]).astype(np.uint32), ids[:600], np.concatenate([[0], toks[5000:5031]]).astype(np.uint32), ids]
    coded = [cz.ac_encode_pairs(*p) for p in g.encode_tokens_batch(streams, 512, 512)]
    assert coded[0][0] == payload and coded[4][0] == payload
    outs = g.decode_tokens_batch([p for p, _ in coded], [s.size - 1 for s in streams], 512, 512, 0)      # 5 streams, groups of 2 -> three lanes
    assert np.array_equal(outs[0], o.decode_tokens(payload, ids.size - 1, 512, 512, 0))                 # same ids as the CPU decoder, decodable or not
    for s_, out, (_, z) in zip(streams, outs, coded):
        if z == 0:                                                                                      # (a zero-width interval makes a stream undecodable)
            assert np.array_equal(out, s_)
    assert sum(1 for _, z in coded if z == 0) >= 3
    g.close(); o.close()


def test_canterbury_through_the_rwkv7_path(cz, czo, golden):
    """BASELINE config 4 (--backend rwkv7 on Canterbury, recurrent-state kernels): alice29 / asyoulik / fields.c -> World symbols -> lock-step
    RWKV7-0.1B-shaped encode -> payloads -> lock-step decode -> bytes; the first chunk's leading symbols are checked against the oracle"""
    wt = cz.WorldTokenizer(os.path.join(golden, "world_vocab_test.json"))
    g = cz.Session("rwkv7", cz.rwkv7_0b1_config(), seed=3, embed_std=0.02, max_streams=16)
    g.debug_set("max_lanes", 2)
    V = g.vocab_size()
    chunks = []
    for name in ("alice29.txt", "asyoulik.txt", "fields.c"):
        data = open(os.path.join(golden, "inputs", name), "rb").read()
        chunks += [data[o:o + 8192] for o in range(0, len(data), 8192)][:7]                 # 7 + 7 + 2 chunks of 8 KiB
    chunks.append(bytes(range(256)) * 2 + b"\xff\xfe tail")                                     # not in the vocabulary: every byte a literal symbol
    streams = [np.concatenate([[0], wt.plan_symbols(c, V)]).astype(np.uint32) for c in chunks]
    assert streams[-1][1:].min() >= V and max(s[1:].max() for s in streams[:-1]) < V
    pairs = g.encode_tokens_batch(streams)
    pays = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
    assert all(zw == 0 for _, zw in pays)
    outs = g.decode_tokens_batch([p for p, _ in pays], [s.size - 1 for s in streams])
    for c, s_, out in zip(chunks, streams, outs):
        assert np.array_equal(out, s_) and wt.detok(out[1:], V) == c
    assert g.debug_get("cdf_fallback_rows") == 0
    # oracle parity on real symbols: the first 320 symbols of alice29's first chunk and the literal-only chunk's first 40
    o = czo.Rwkv7Session(czo.rwkv7_0b1_config(), seed=3, embed_std=0.02)
    for k, n in ((0, 320), (len(chunks) - 1, 40)):
        ref_p, clo, chi, _ = o.encode_tokens(streams[k][: n + 1])
        assert np.array_equal(pairs[k][0][:n], clo) and np.array_equal(pairs[k][1][:n], chi), k
    g.close(); o.close(); wt.close()


def _corpus(*argv, env=None):
    r = subprocess.run([sys.executable, "-m", "candlezip_amd.corpus", *argv], cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_corpus_driver_rwkv7_backend(golden):
    files = [os.path.join(golden, "inputs", n) for n in ("fields.c", "asyoulik.txt")]
    d = _corpus("--backend", "rwkv7", "--synthetic", "5", "--rwkv-tokenizer", os.path.join(golden, "world_vocab_test.json"), "--chunk-bytes", "16384",
                "--max-streams", "16", *files)
    assert d["backend"] == "rwkv7" and d["roundtrip_failures"] == [] and d["chunks"] == 1 + 8 and d["lossy_utf8_chunks"] == 0 and d["tokens"] > 70000


def _eq(a, b, what):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32); b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    n = int(np.sum(a != b))
    assert n == 0, f"{what}: {n}/{a.size} logits differ"


@pytest.mark.parametrize("path", ["tiled", "np2", "hybrid", "mid", "mid_q0", "mid8", "mid9", "gemv"])
@pytest.mark.parametrize("shape", ["tiny", "135m"])
def test_wide_group_step_is_bitexact(cz, czo, shape, path):
    """the decode step of ONE WIDE lock-step group against the oracle: every logit of the trait steps, the coder intervals, and a lock-step
    decode across a re-prime, graph replay and eager.  Two forms: "tiled" = tiled MFMA GEMMs with 64-row tiles + rope kernel + dec_attn with
    natural-layout output; "np2" = dec_gemm_kernel with 32-row x 64-column blocks (MT = 2, NP = 2), the shape groups of >= 256 streams take;
    "hybrid" = tiled QKV / lm_head projections, the rest on the 32 x 64 blocks; "mid" = gemm_mid_kernel (32 x 32 tiles, four K groups per
    block, ordered hand-off of the chunk partials, RoPE + cache write in the QKV epilogue) for the layer projections — the four waves sharing
    the ordered sum and the epilogue by register quads ("mid", round 3) or wave 0 doing both ("mid_q0"); "mid8" / "mid9" = the
    same with eight K groups / one wave per 64-chunk (nine for K = 576, twelve in two rounds for K = 1536): the `mid_nw` tuning knob; "gemv" (round 4) =
    the VALU projections groups of <= 4 streams take by default (kernels_gemv.hip: one wave per block, fused norms): the trait steps and lock-step
    decodes of 1, 2, 3 and 4 streams — every other path runs with them switched off (`gemv_max_rows` = 0), so that its own kernels serve the single-stream steps too"""
    if shape == "tiny":
        gcfg, ocfg, V, n_tok, seed = cz.tiny_llama_config(), czo.tiny_llama_config(), 512, 700, 11
    else:
        gcfg, ocfg, V, n_tok, seed = cz.smollm2_135m_config(), czo.smollm2_135m_config(), 49152, 40, 0x5EED
    g = cz.Session("smollm", gcfg, seed=seed, embed_std=0.05 if shape == "tiny" else 0.02, max_streams=6)
    o = czo.SmolLmSession(ocfg, seed=seed, embed_std=0.05 if shape == "tiny" else 0.02)
    g.debug_set("gemv_max_rows", 16 if path == "gemv" else 0)                 # ("gemv": the shipped value — the five-stream decode below runs as row groups of 4 + 1)
    g.debug_set("dec_attn_split_max", 0 if path in ("tiled", "np2", "hybrid", "mid8") else 128)   # small groups: the one-kernel forms below (0) / the three-launch split form (round 3)
    g.debug_set("dec_attn_small_grid", 0 if path in ("tiled", "np2") else 128)   # decode attention with 8 waves per block (whole-device grids) / 16 (small grids)
    g.debug_set("dec_attn_split_heads", 0 if path == "hybrid" else 1)            # small grids: a block per kv head (0) / per query head (1, the default)
    if path == "tiled":
        g.debug_set("wide_min_streams", 1)                                # every group, even one stream, takes the tiled-GEMM step
        g.debug_set("no_last_prune", 1)                                   # ... and the last layer runs its tail on every row (the other paths: wanted rows only)
    elif path.startswith("mid"):                                          # the K-group GEMM (kernels_gemm_mid.hip) for the four layer projections
        g.debug_set("wide_min_streams", 1); g.debug_set("wide_mode", 3)
        g.debug_set("mid_nw", {"mid": 4, "mid_q0": 4, "mid8": 8, "mid9": 0}[path])     # (process-wide knob: reset by conftest)
        g.debug_set("mid_quad", 0 if path == "mid_q0" else 1)             # four-wave blocks: wave 0 alone sums and runs the epilogue (0) / all four waves share both (1, the default)
    elif path == "gemv":
        pass                                                              # the shipped narrow-group configuration
    elif path == "hybrid":                                                # tiled QKV / lm_head, dec_gemm 32 x 64 blocks for the other projections
        g.debug_set("wide_min_streams", 1); g.debug_set("wide_mode", 2); g.debug_set("dec_wide_rows", 1)
    else:
        g.debug_set("wide_min_streams", 0); g.debug_set("dec_wide_rows", 1)   # ... or the wide dec_gemm blocks (process-wide knob: reset by conftest)
    rng = np.random.default_rng(21)
    toks = rng.integers(0, V, 30).astype(np.uint32)
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:1]), o.reprime_logits(toks[:1]), "bos")
    for i, t in enumerate(toks[1:8]):
        _eq(g.step_logits_tensor(t), o.step_logits(t), f"wide step {i}")
    _eq(g.reprime_with_history_and_get_last_logits_tensor(toks[:20]), o.reprime_logits(toks[:20]), "reprime(20)")
    _eq(g.step_logits_tensor(toks[20]), o.step_logits(toks[20]), "wide step after reprime")
    lens = [n_tok, n_tok // 2, 3, n_tok, 1]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    ref = [o.encode_tokens(s_) for s_ in streams[:2]]
    pairs = g.encode_tokens_batch(streams)
    for (lo, hi), (rp, clo, chi, _) in zip(pairs, ref):
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    coded = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
    assert coded[0][0] == ref[0][0]
    for no_graph in (0, 1):
        g.debug_set("no_graph", no_graph)
        outs = g.decode_tokens_batch([p for p, _ in coded], lens)
        assert np.array_equal(outs[0], o.decode_tokens(ref[0][0], lens[0]))
        for s_, out, (_, z) in zip(streams, outs, coded):
            if z == 0:
                assert np.array_equal(out, s_)
        if path == "gemv":                                                # groups of 1..4 streams: the one-, two- and four-row instantiations (three rows: a ragged four)
            for nb in (1, 2, 3, 4):
                g.debug_set("gemv_fold_attn", nb & 1)                      # the split attention's ordered sum inside the o-projection (the default) / as its own launch
                outs = g.decode_tokens_batch([p for p, _ in coded[:nb]], lens[:nb])
                assert np.array_equal(outs[0], o.decode_tokens(ref[0][0], lens[0])), nb
                for s_, out, (_, z) in zip(streams[:nb], outs, coded[:nb]):
                    if z == 0:
                        assert np.array_equal(out, s_), nb
    # (the process-wide knobs touched above go back to their shipped values in conftest's fixture finalizer)
    g.close(); o.close()


@pytest.mark.parametrize("backend", ["smollm", "rwkv7"])
def test_decode_with_payloads_fed_in_pieces(cz, backend):
    """cz_decode_begin_partial_batch / cz_decode_feed_batch: the encoder runs ahead of the decoder window by window (cz_ac_encoder_*), the
    decoder only ever sees the bytes that are final; ids equal the one-shot decode.  Advancing past what the fed bytes allow is reported."""
    if backend == "smollm":
        g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=2)
        V = 512
    else:
        g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=2)
        V = 512 + 256
    g.debug_set("max_lanes", 2)
    rng = np.random.default_rng(31)
    lens = [1300, 900, 1300]                                                       # two groups on two lanes
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    W, LAG = 256, 2
    encs = [cz.AcEncoder() for _ in streams]
    pieces = [[] for _ in streams]
    nwin = -(-max(lens) // W)

    def compress_window(w):
        for k, (lo, hi) in enumerate(pairs):
            b, e = w * W, min((w + 1) * W, lens[k])
            if b < lens[k]:
                pieces[k].append(encs[k].push(lo[b:e], hi[b:e]))
                if e == lens[k]:
                    pieces[k][-1] += encs[k].finish()[0]
            else:
                pieces[k].append(b"")
    for w in range(LAG):
        compress_window(w)
    first = [b"".join(p) for p in pieces]
    g.decode_begin_partial_batch(first, [4 * n + 16 for n in lens], lens)
    for w in range(nwin):
        if w + LAG < nwin:
            compress_window(w + LAG)
            g.decode_feed_batch([p[w + LAG] for p in pieces], final=(w + LAG == nwin - 1))
        g.decode_advance(W)
    outs = g.decode_fetch_batch()
    g.decode_end()
    for s_, o in zip(streams, outs):
        assert np.array_equal(s_, o)
    assert [b"".join(p) for p in pieces] == [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    # starved decoder: only the first window's bytes, yet all tokens requested -> reported, not silent
    g.decode_begin_partial_batch([p[0] for p in pieces], [4 * n + 16 for n in lens], lens)
    g.decode_advance(max(lens))
    with pytest.raises(cz.CandlezipError, match="fed so far"):
        g.decode_fetch_batch()
    g.decode_end()
    with pytest.raises(cz.CandlezipError):
        g.decode_begin_batch(first, lens); g.decode_feed_batch([b"x"] * 3)        # a decode begun with complete payloads takes no feed
    g.decode_end()
    g.close()


def test_full_size_samples_round_trip_and_windowed_encode_agree(cz):
    """BASELINE config 3's per-stream size (a 64 KiB enwik8 chunk = 17 408 tokens, 34 re-prime segments) at the SmolLM2-135M shape, where the
    oracle would need hours: size-independent properties instead.  (1) the window-by-window range encode + incremental coder (the bench's
    compress stage) gives byte for byte the payload of the one-shot encode; (2) groups of 3 + 2 streams on two lanes decode the input back,
    34 re-primes each; (3) the payload of a stream does not depend on which other streams share its batch."""
    import hashlib
    T, W = 17408, 512
    g = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=3)
    rng = np.random.default_rng(2024)
    streams = [np.concatenate([[0], rng.integers(0, 49152, T)]).astype(np.uint32) for _ in range(5)]
    pairs = g.encode_tokens_batch(streams, 512, 512)
    payloads = [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in pairs]
    # (1) windows of 512 tokens, coder fed window by window
    encs = [cz.AcEncoder() for _ in streams]
    pieces = [b""] * len(streams)
    for w in range(T // W):
        part = g.encode_tokens_range_batch(streams, [(w * W, (w + 1) * W)] * len(streams), 512, 512)
        for k, (lo, hi) in enumerate(part):
            pieces[k] += encs[k].push(lo, hi)
    for k in range(len(streams)):
        pieces[k] += encs[k].finish()[0]
        encs[k].close()
    assert [hashlib.sha256(p).hexdigest() for p in pieces] == [hashlib.sha256(p).hexdigest() for p in payloads]
    # (3) one stream alone
    alone = g.encode_tokens_batch(streams[3:4], 512, 512)[0]
    assert np.array_equal(alone[0], pairs[3][0]) and np.array_equal(alone[1], pairs[3][1])
    # (2) decode: groups of 3 + 2 on two lanes
    g.debug_set("max_lanes", 2)
    outs2 = g.decode_tokens_batch(payloads, [T] * len(streams), 512, 512, 0)
    for s_, o in zip(streams, outs2):
        assert np.array_equal(s_, o)
    g.close()

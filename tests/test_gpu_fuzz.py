"""Randomised schedules against the oracle (tiny model): odd contexts and re-prime intervals, ragged and degenerate streams, lanes on
and off.  Every case checks the coder intervals of every stream bit for bit and the lock-step decode round trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def test_random_schedules_match_oracle(cz, czo):
    import os
    rng = np.random.default_rng(int(os.environ.get("CZ_FUZZ_SEED", "20261004")))       # (a soak run: CZ_FUZZ_SEED=<other> CZ_FUZZ_CASES=<more>)
    cfg_g, cfg_o = cz.tiny_llama_config(), czo.tiny_llama_config()
    o = czo.SmolLmSession(cfg_o, seed=11, embed_std=0.05)
    sessions = {}
    checked = 0
    for case in range(int(os.environ.get("CZ_FUZZ_CASES", "36"))):
        context = int(rng.choice([2, 3, 17, 64, 100, 255, 256, 512, 513, 700, 1024]))
        reprime = int(rng.choice([1, 2, 7, 63, 64, 65, 200, 511, 512, 513, 1000, 1536, 1537]))
        # the KV sawtooth peaks at min(context, 511) + reprime: the tiny model's max_position_embeddings (2048) is the only limit, as in the
        # reference (511 + 1537 = 2048 positions exactly: the last schedule that fits)
        n_streams = int(rng.integers(1, 9))
        max_streams = int(rng.choice([1, 2, 3, 8]))
        lanes = int(rng.choice([1, 2, 3]))
        lens = [int(x) for x in rng.choice([0, 1, 2, 5, 63, 64, 65, 130, 300, 600], n_streams)]
        if reprime >= 1000 or case % 9 == 4:
            lens[0] = int(rng.choice([1600, 2100, 2300]))         # long enough to run the sawtooth up to its peak
        streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
        key = max_streams
        if key not in sessions:
            sessions[key] = cz.Session("smollm", cfg_g, seed=11, embed_std=0.05, max_streams=max_streams)
        g = sessions[key]
        g.debug_set("max_lanes", lanes)
        pairs = g.encode_tokens_batch(streams, context, reprime)
        pays = []
        for s, (lo, hi) in zip(streams, pairs):
            ref_p, clo, chi, zw = o.encode_tokens(s, context, reprime)
            assert np.array_equal(lo, clo) and np.array_equal(hi, chi), (case, context, reprime, s.size)
            p, zw2 = cz.ac_encode_pairs(lo, hi)
            assert p == ref_p and zw == zw2, (case, context, reprime)
            pays.append(p)
            checked += s.size - 1
        if any(cz.ac_encode_pairs(lo, hi)[1] for lo, hi in pairs):
            continue                                              # a zero-width interval (H3): the reference itself cannot decode it
        outs = g.decode_tokens_batch(pays, [s.size - 1 for s in streams], context, reprime, 0)
        # decode uses min(context - 1, 511) where encode used min(context, 511) (src/main.rs:1935 vs :2506): the two schedules agree,
        # and the stream round-trips, only when both clamp to 511 — otherwise compare with the oracle's own decoder instead
        for s, p, out in zip(streams, pays, outs):
            ref = o.decode_tokens(p, s.size - 1, context, reprime, 0)
            assert np.array_equal(out, ref), (case, context, reprime, s.size)
            if context >= 512:
                assert np.array_equal(out, s), (case, context, reprime)
        assert g.debug_get("cdf_fallback_rows") == 0
    for g in sessions.values():
        g.close()
    o.close()
    assert checked > 5000


def test_random_rwkv_batches_match_oracle(cz, czo):
    """random batches through every RWKV7 form: token-by-token and sequence-parallel encode (time blocks of 7 / 64 / 32 768 rows), the narrow and the
    wide (grouped-GEMM) decode step, lanes, literal symbols; and random spans through the span scorer.  CZ_FUZZ_RWKV_SEED / CZ_FUZZ_RWKV_CASES: soak runs."""
    import os
    rng = np.random.default_rng(int(os.environ.get("CZ_FUZZ_RWKV_SEED", "77")))
    o = czo.Rwkv7Session(czo.tiny_rwkv7_config(), seed=9, embed_std=0.05)
    for case in range(int(os.environ.get("CZ_FUZZ_RWKV_CASES", "10"))):
        n_streams = int(rng.integers(1, 8)); max_streams = int(rng.choice([1, 2, 4])); lanes = int(rng.choice([1, 2, 3]))
        seq = int(rng.choice([0, 1, 1])); seq_rows = int(rng.choice([7, 64, 32768])); wide = int(rng.choice([0, 1]))
        lens = [int(x) for x in rng.choice([0, 1, 2, 17, 64, 90, 150], n_streams)]
        lit = float(rng.choice([0.0, 0.05, 0.5]))                 # share of literal-byte symbols (they are coded but do not step the model)
        streams = []
        for n in lens:
            body = rng.integers(0, 512, n)
            mask = rng.random(n) < lit
            body[mask] = 512 + rng.integers(0, 256, int(mask.sum()))
            streams.append(np.concatenate([[0], body]).astype(np.uint32))
        g = cz.Session("rwkv7", cz.tiny_rwkv7_config(), seed=9, embed_std=0.05, max_streams=max_streams)
        g.debug_set("max_lanes", lanes)
        g.debug_set("rwkv_seq", seq); g.debug_set("rwkv_seq_rows", seq_rows)
        if wide:
            g.debug_set("rwkv_wide_min", 1)                      # the wide (grouped-GEMM) decode step even for these few streams
        pairs = g.encode_tokens_batch(streams)
        pays = []
        for s, (lo, hi) in zip(streams, pairs):
            ref_p, clo, chi, zw = o.encode_tokens(s)
            assert np.array_equal(lo, clo) and np.array_equal(hi, chi), (case, s.size, seq, seq_rows)
            pays.append(cz.ac_encode_pairs(lo, hi)[0])
            assert pays[-1] == ref_p
        outs = g.decode_tokens_batch(pays, [s.size - 1 for s in streams])
        for s, out in zip(streams, outs):
            assert np.array_equal(out, s), (case, wide, lanes)
        # the span scorer on pieces of the same streams: history with literals (filtered), a hint now and then, literal and out-of-range targets
        spans = []
        for s in streams:
            if s.size < 4:
                continue
            cut = int(rng.integers(1, s.size - 1))
            tg = s[cut: cut + int(rng.integers(1, 40))].copy()
            if rng.random() < 0.3:
                tg[int(rng.integers(0, tg.size))] = 512 + 256 + int(rng.integers(0, 1000))        # past vocab + 256: ac_p_min(), no step
            hint = rng.integers(0, 512, int(rng.integers(0, 6))) if rng.random() < 0.5 else ()
            if not (np.any(s[:cut] < 512) or len(hint)):
                continue                                                                       # (an empty prime is the reference's error; covered in test_gpu_model.py)
            spans.append((s[:cut], tg, hint))
        if spans:
            want = [o.xe_bits_over_span(h, t, hint)[0] for h, t, hint in spans]
            got = g.xe_bits_batch([x[0] for x in spans], [x[1] for x in spans], [x[2] for x in spans])
            assert got.tolist() == want, case
        assert g.debug_get("cdf_fallback_rows") == 0
        g.close()
    o.close()

"""A-1 / A-1b / A-6 / A-7 / R-1: oracle model sessions — internal consistency (the properties the HIP path relies on)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def tiny(czo):
    s = czo.SmolLmSession(czo.tiny_llama_config(), seed=11, embed_std=0.05)
    yield s
    s.close()


def test_step_equals_batched_bitwise(czo, tiny):
    rng = np.random.default_rng(0)
    toks = rng.integers(0, tiny.V, 150).astype(np.uint32)
    tiny.reprime_logits(toks[:1])
    steps = [tiny.reprime_logits(toks[:1])] + [tiny.step_logits(t) for t in toks[1:]]
    rows = tiny.forward_rows(toks)
    assert np.array_equal(np.array(steps).view(np.uint32), rows.view(np.uint32))
    assert tiny.index_pos() == 150
    # reprime = last row of the batched forward, and then stepping continues from position n
    last = tiny.reprime_logits(toks[:77])
    assert np.array_equal(last.view(np.uint32), rows[76].view(np.uint32)) and tiny.index_pos() == 77
    nxt = tiny.step_logits(toks[77])
    assert np.array_equal(nxt.view(np.uint32), rows[77].view(np.uint32))


def test_encode_decode_roundtrip_with_reprimes(czo, tiny):
    rng = np.random.default_rng(1)
    n = 1300                                                   # > 2 re-prime boundaries at ctx 512 / interval 512
    ids = np.concatenate([[0], rng.integers(0, tiny.V, n)]).astype(np.uint32)
    payload, clo, chi, zw = tiny.encode_tokens(ids)
    assert zw == 0 and np.all(chi > clo)
    # A-4 over the emitted pairs reproduces the payload (this is how the GPU encode path assembles it)
    again, zw2 = czo.encode_pairs(clo, chi)
    assert again == payload
    out = tiny.decode_tokens(payload, n)
    assert np.array_equal(out, ids)
    # random symbols under a near-uniform synthetic model cost >= log2(V) bits each (cross-entropy), not much more
    assert np.log2(tiny.V) <= 8 * len(payload) / n < np.log2(tiny.V) + 2


def test_reprime_schedule_matches_reference_predicate(czo, tiny):
    """src/main.rs:2280: reprime iff index_pos >= min(ctx,511) and i % R == 0 and i > 0; window = last 511 ids."""
    rng = np.random.default_rng(2)
    ids = np.concatenate([[0], rng.integers(0, tiny.V, 600)]).astype(np.uint32)
    _, clo, chi, _ = tiny.encode_tokens(ids, context=512, reprime_interval=512)
    # replay by hand
    lg = tiny.reprime_logits(ids[:1])
    for i in range(600):
        if tiny.index_pos() >= 511 and i % 512 == 0 and i > 0:
            assert i == 512
            lg = tiny.reprime_logits(ids[i + 1 - 511:i + 1]); assert tiny.index_pos() == 511
        cdf = czo.logits_to_cdf(lg)
        assert (cdf[ids[i + 1]], cdf[ids[i + 1] + 1]) == (clo[i], chi[i])
        lg = tiny.step_logits(ids[i + 1])


def test_context_1024_window_is_still_511(czo, tiny):
    rng = np.random.default_rng(3)
    ids = np.concatenate([[0], rng.integers(0, tiny.V, 1100)]).astype(np.uint32)
    p, clo, chi, _ = tiny.encode_tokens(ids, context=1024, reprime_interval=1024)
    # first reprime at i = 1024 with the last 511 ids (min(1024, 511), src/main.rs:1935)
    lg = tiny.reprime_logits(ids[1025 - 511:1025])
    cdf = czo.logits_to_cdf(lg)
    assert (cdf[ids[1025]], cdf[ids[1025] + 1]) == (clo[1024], chi[1024])
    assert np.array_equal(tiny.decode_tokens(p, 1100, context_window=1024, reprime_interval=1024), ids)


def test_peaked_model_hits_zero_width_and_reports_it(czo):
    s = czo.SmolLmSession(czo.tiny_llama_config(), seed=5, embed_std=1.5)      # logits std ~ 17 => p < 2^-30 exists
    rng = np.random.default_rng(4)
    ids = np.concatenate([[0], rng.integers(0, s.V, 200)]).astype(np.uint32)
    payload, clo, chi, zw = s.encode_tokens(ids)
    assert zw == int(np.sum(clo == chi)) and zw > 0                            # H3: counted, not fixed
    s.close()


def test_synth_generator_is_pure_function(czo):
    L = czo.lib()
    a = [L.czo_synth_bf16(1, 2, i, 0.02, 0.0) for i in range(2000)]
    b = [L.czo_synth_bf16(1, 2, i, 0.02, 0.0) for i in range(2000)]
    assert a == b and len(set(a)) > 500
    v = (np.array(a, np.uint32) << 16).view(np.float32)
    assert abs(v.std() - 0.02) < 2e-3 and abs(v.mean()) < 2e-3


def test_rwkv7_roundtrip_and_state_reset(czo):
    s = czo.Rwkv7Session(czo.tiny_rwkv7_config(), seed=9, embed_std=0.05)
    rng = np.random.default_rng(5)
    ids = np.concatenate([[0], rng.integers(0, s.V, 300)]).astype(np.uint32)
    a = [s.step_logits(t) for t in ids[:20]]
    s.reset()
    b = [s.step_logits(t) for t in ids[:20]]
    assert np.array_equal(np.array(a).view(np.uint32), np.array(b).view(np.uint32))
    assert np.isfinite(np.array(a)).all() and np.array(a).std() > 0
    payload, clo, chi, zw = s.encode_tokens(ids)
    assert zw == 0
    assert np.array_equal(s.decode_tokens(payload, 300), ids)
    # literal symbols (vocab + byte) are codable and do not step the model (src/main.rs:2347)
    ids2 = ids.copy(); ids2[5] = s.V + 65; ids2[9] = s.V + 255
    p2, _, _, _ = s.encode_tokens(ids2)
    assert np.array_equal(s.decode_tokens(p2, 300), ids2)
    s.close()

"""Round-4 GPU parity tests: decode-CDF blocks that own several streams (the row loop of the DECODE instantiations of cdf_par_kernel, which
real launches never take: one block per stream), the shipped values of the process-wide knobs, and a KV cache that is replaced (not grown)
between two runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


@pytest.mark.parametrize("backend,shape", [("rwkv7", "0b1"), ("rwkv7", "tiny"), ("smollm", "tiny"), ("smollm", "135m")])
def test_decode_blocks_that_own_several_streams(cz, czo, backend, shape):
    """cdf_dec_grid_cap = 2: five streams on two blocks, so a block walks rows 0, 2, 4 / 1, 3 of ONE launch — the staged row load of the
    literals mode (RWKV7-0.1B alphabet: 68 logits per thread) overlays LDS fields the previous row reads until its end; finished and empty
    streams take the early `continue`.  Against the oracle's payloads, graph replay and eager."""
    if backend == "rwkv7":
        if shape == "tiny":
            gcfg, ocfg, V, n_tok, seed, std = cz.tiny_rwkv7_config(), czo.tiny_rwkv7_config(), 512, 90, 9, 0.05
        else:
            gcfg, ocfg, V, n_tok, seed, std = cz.rwkv7_0b1_config(), czo.rwkv7_0b1_config(), 65536, 9, 3, 0.02
        o = czo.Rwkv7Session(ocfg, seed=seed, embed_std=std)
    else:
        if shape == "tiny":
            gcfg, ocfg, V, n_tok, seed, std = cz.tiny_llama_config(), czo.tiny_llama_config(), 512, 90, 11, 0.05
        else:
            gcfg, ocfg, V, n_tok, seed, std = cz.smollm2_135m_config(), czo.smollm2_135m_config(), 49152, 9, 0x5EED, 0.02
        o = czo.SmolLmSession(ocfg, seed=seed, embed_std=std)
    g = cz.Session(backend, gcfg, seed=seed, embed_std=std, max_streams=5)
    rng = np.random.default_rng(41)
    lens = [n_tok, n_tok // 2, 1, 0, n_tok - 2]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    if backend == "rwkv7":
        streams[0][2] = V + 65; streams[4][3] = V + 200                     # literal symbols: coded, never stepped
    pays = []
    for s_, (lo, hi) in zip(streams, g.encode_tokens_batch(streams)):
        ref_p, clo, chi, zw = o.encode_tokens(s_)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
        pays.append((ref_p, zw))
    for cap in (2, 0):
        g.debug_set("cdf_dec_grid_cap", cap)
        for no_graph in (0, 1):
            g.debug_set("no_graph", no_graph)
            outs = g.decode_tokens_batch([p for p, _ in pays], lens)
            for s_, out, (_, zw) in zip(streams, outs, pays):
                if zw == 0:
                    assert np.array_equal(out, s_), (cap, no_graph)
    assert g.debug_get("cdf_fallback_rows") == 0
    g.close(); o.close()


def test_process_knobs_go_back_to_their_shipped_values(cz):
    """every process-wide knob is restored by cz_debug_set("reset_process_knobs") (conftest's finalizer after each GPU test): a session that
    turned them decodes like a fresh process afterwards"""
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=4)
    rng = np.random.default_rng(5)
    ids = np.concatenate([[0], rng.integers(0, 512, 200)]).astype(np.uint32)
    lo, hi = g.encode_tokens(ids)
    pay, zw = cz.ac_encode_pairs(lo, hi)
    shipped = {k: g.debug_get(k) for k in ("dec_attn_split_max", "mid_nw", "mid_quad", "dec_wide_rows", "cdf_dec_grid_cap")}
    assert shipped["dec_attn_split_max"] == 36 and shipped["mid_nw"] == 4 and shipped["cdf_dec_grid_cap"] == 0
    for k in shipped:
        g.debug_set(k, 1)
    assert all(g.debug_get(k) == 1 for k in shipped)
    cz.reset_process_knobs()
    assert {k: g.debug_get(k) for k in shipped} == shipped
    if zw == 0:
        assert np.array_equal(g.decode_tokens(pay, 200), ids)
    g.close()


def test_kv_cache_replaced_between_runs_keeps_results(cz, czo):
    """a run whose schedule needs a longer KV cache than the previous run left behind: nothing of the old cache is kept, so it is given back
    BEFORE the longer one is allocated (ensure_streams); the new run and a later short run decode exactly"""
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=2)
    o = czo.SmolLmSession(czo.tiny_llama_config(), seed=11, embed_std=0.05)
    g.debug_set("kv_ctx_cap", 256)                                           # a deliberately short first cache
    rng = np.random.default_rng(2)
    short = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in (120, 60)]
    long_ = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in (700, 300)]
    for streams, (ctx, R) in ((short, (512, 512)), (long_, (512, 2048)), (short, (512, 512))):     # 121 keys; 701 keys in one sawtooth tooth; 121 again
        pairs = g.encode_tokens_batch(streams, ctx, R)
        pays = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
        ref_p, clo, chi, zw = o.encode_tokens(streams[0], ctx, R)
        assert pays[0][0] == ref_p
        outs = g.decode_tokens_batch([p for p, _ in pays], [len(s_) - 1 for s_ in streams], ctx, R)
        for s_, out, (_, z) in zip(streams, outs, pays):
            if z == 0:
                assert np.array_equal(out, s_)
    g.close(); o.close()


@pytest.mark.parametrize("shape", ["tiny", "0b1"])
@pytest.mark.parametrize("gemv", [16, 0])
def test_rwkv7_narrow_groups_valu_projections_are_bitexact(cz, czo, shape, gemv):
    """RWKV7 groups of 1..4 streams on the one-wave VALU projections (kernels_gemv.hip: grouped launch of the time-mix's seven projections and of
    its four LoRA up-projections, relu^2 key projection, K = 3072 value projection, masked lm_head) — `gemv_max_rows` = 16, the shipped value — and
    on dec_gemm_kernel (0): trait-step logits, the lock-step encode walk and decodes with literal symbols against the oracle, graph and eager"""
    if shape == "tiny":
        gcfg, ocfg, V, n_tok, seed, std = cz.tiny_rwkv7_config(), czo.tiny_rwkv7_config(), 512, 120, 9, 0.05
    else:
        gcfg, ocfg, V, n_tok, seed, std = cz.rwkv7_0b1_config(), czo.rwkv7_0b1_config(), 65536, 9, 3, 0.02
    g = cz.Session("rwkv7", gcfg, seed=seed, embed_std=std, max_streams=6)
    o = czo.Rwkv7Session(ocfg, seed=seed, embed_std=std)
    g.debug_set("gemv_max_rows", gemv); g.debug_set("rwkv_seq", 0)          # (rwkv_seq 0: the encode walks the recurrence token by token through the same step)
    rng = np.random.default_rng(18)
    toks = rng.integers(0, V, 10).astype(np.uint32)
    o.reset()
    a = g.reprime_with_history_and_get_last_logits_tensor(toks[:1]); b = o.step_logits(toks[0])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for t in toks[1:6]:
        a = g.step_logits_tensor(t); b = o.step_logits(t)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    lens = [n_tok, n_tok // 2, 2, n_tok - 3, n_tok // 3, 1]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    streams[0][2] = V + 65; streams[0][3] = V + 1; streams[3][n_tok - 4] = V + 200
    pays = []
    for s_, (lo, hi) in zip(streams, g.encode_tokens_batch(streams)):
        ref_p, clo, chi, zw = o.encode_tokens(s_)
        assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
        pays.append(ref_p)
    for nb in (1, 2, 3, 4, 6):                                               # (six: past the four streams RWKV7 groups take the VALU projections for)
        for no_graph in (0, 1):
            g.debug_set("no_graph", no_graph)
            for s_, out in zip(streams[:nb], g.decode_tokens_batch(pays[:nb], lens[:nb])):
                assert np.array_equal(out, s_), (nb, no_graph)
    g.close(); o.close()


@pytest.mark.parametrize("backend,n_streams", [("smollm", 150), ("rwkv7", 84)])
def test_groups_at_the_wide_step_thresholds_round_trip(cz, czo, backend, n_streams):
    """round 4 moved the wide decode step down to groups of 144 (SmolLM) / 80 (RWKV7) streams: ONE group just past the threshold — a row count that is no
    multiple of the 32-row tiles — with the shipped knobs, at the BASELINE shapes: stream 0 against the oracle, every stream's round trip, and the same
    payloads decoded in groups of 12 (the narrow forms) give the same ids"""
    if backend == "smollm":
        gcfg, ocfg, V, n_tok, seed, std = cz.smollm2_135m_config(), czo.smollm2_135m_config(), 49152, 14, 0x5EED, 0.02
        o = czo.SmolLmSession(ocfg, seed=seed, embed_std=std)
    else:
        gcfg, ocfg, V, n_tok, seed, std = cz.rwkv7_0b1_config(), czo.rwkv7_0b1_config(), 65536, 8, 3, 0.02
        o = czo.Rwkv7Session(ocfg, seed=seed, embed_std=std)
    g = cz.Session(backend, gcfg, seed=seed, embed_std=std, max_streams=n_streams)
    rng = np.random.default_rng(77)
    lens = [n_tok - (k % 3) for k in range(n_streams)]
    streams = [np.concatenate([[0], rng.integers(0, V, n)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    coded = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
    ref_p = o.encode_tokens(streams[0])[0]
    assert coded[0][0] == ref_p
    outs = g.decode_tokens_batch([p for p, _ in coded], lens)                  # one group of n_streams: the wide step
    for s_, out, (_, z) in zip(streams, outs, coded):
        if z == 0:
            assert np.array_equal(out, s_)
    g.close()
    g2 = cz.Session(backend, gcfg, seed=seed, embed_std=std, max_streams=12)   # groups of 12 on lanes: the narrow forms
    g2.debug_set("max_lanes", 2)
    outs2 = g2.decode_tokens_batch([p for p, _ in coded[:36]], lens[:36])
    for a, b in zip(outs[:36], outs2):
        assert np.array_equal(a, b)
    g2.close(); o.close()


def test_device_expf_equals_its_definition_on_every_float(cz):
    """round 4: the device form of cz_expf (clamps as v_max / v_min, ONE v_ldexp_f32 instead of the two-step 2^n scaling; cz_numerics.h) against the definition
    (cz_expf_ref, the form the host code and the oracle's restatement use) on ALL 2^32 float bit patterns — NaNs, infinities, subnormal results included —
    on the device: not one differing bit."""
    import ctypes as C
    out = (C.c_ulonglong * 3)()
    assert cz.lib().cz_debug_expf_exhaustive(out) == 0
    assert out[2] == 1 << 32
    assert out[0] == 0, "device cz_expf differs from its definition on %d inputs, the first: bits 0x%08x" % (out[0], out[1])


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (300, 192, 128), (129, 256, 192), (512, 576, 576), (1000, 3072, 576)])
def test_big_gemm_half_tile_pipeline_equals_one_buffer_loop_and_oracle(cz, czo, M, N, K):
    """round 4: the 128-row tiles of the big-M GEMM stage half K tiles (32 k, two swizzled LDS buffers) between the wave's own MFMAs (`gemm_pipe` 3, shipped) —
    one and two K tiles (K = 64, 128: the pipeline's prologue and its last half), ragged last row tiles, both tile widths (N % 128 == 0 -> 128 x 128, else 128 x 64),
    with and without a residual: bit for bit the one-buffer loop (`gemm_pipe` 0) and the oracle's k-ascending chain"""
    rng = np.random.default_rng(M + N + K)
    A = (rng.standard_normal((M, K)) * rng.choice([1e-3, 1.0, 30.0], (M, 1))).astype(np.float32)
    W = czo.to_bf16(rng.standard_normal((N, K)) * 0.05)
    R = rng.standard_normal((M, N)).astype(np.float32)
    g = cz.Session("smollm", cz.tiny_llama_config(), seed=11, embed_std=0.05, max_streams=1)      # (a handle for the process-wide knob)
    for res in (None, R):
        ref = czo.gemm_bf16(A, W, res)
        for pipe in (3, 0, 1, 2):
            g.debug_set("gemm_pipe", pipe)
            C = np.empty((M, N), np.float32)
            assert cz.lib().cz_debug_gemm(A.ctypes.data, W.ctypes.data, res.ctypes.data if res is not None else None, C.ctypes.data, M, N, K, 1) == 0
            assert np.array_equal(C.view(np.uint32), ref.view(np.uint32)), (pipe, res is not None)
    g.close()


@pytest.mark.parametrize("waves", [4, 8, 0])
def test_decode_attention_block_shapes_round_trip_at_long_contexts(cz, czo, waves):
    """round 4: the whole-device decode attention with FOUR waves per block (each wave walks up to eight 64-key chunks; picked where lock-step groups share the device) against
    eight: 150 streams of a three-query-heads-per-KV-head model decoded through the sawtooth 512 -> 1023 keys (up to sixteen chunks: two to four per wave), ragged lengths;
    stream 0's payload against the oracle, every stream's round trip, the same ids from either shape"""
    kw = dict(d_model=384, n_heads=6, n_kv_heads=2, d_ffn=512)
    g = cz.Session("smollm", cz.tiny_llama_config(**kw), seed=21, embed_std=0.05, max_streams=150)
    o = czo.SmolLmSession(czo.tiny_llama_config(**kw), seed=21, embed_std=0.05)
    g.debug_set("dec_attn_waves", waves)
    rng = np.random.default_rng(99)
    lens = [1100 - 7 * (k % 11) for k in range(150)]
    streams = [np.concatenate([[0], rng.integers(0, 512, n)]).astype(np.uint32) for n in lens]
    pairs = g.encode_tokens_batch(streams)
    coded = [cz.ac_encode_pairs(lo, hi) for lo, hi in pairs]
    assert coded[0][0] == o.encode_tokens(streams[0])[0]
    outs = g.decode_tokens_batch([p for p, _ in coded], lens)
    for s_, out, (_, z) in zip(streams, outs, coded):
        if z == 0:
            assert np.array_equal(out, s_)
    g.close(); o.close()

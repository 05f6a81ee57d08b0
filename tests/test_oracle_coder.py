"""A-2..A-5: oracle coder + pdf/cdf against hand-derived KATs, the pure-Python restatement and round trips.
The reference has no tests for these functions: KATs are authored from the source text."""
import numpy as np
import pytest
from pyref import PyEncoder, PyDecoder

T = 1 << 30


def test_kat_single_symbol_half(czo):
    # hand-derived: one symbol occupying [0, T/2): range 2^32 -> low=0, high=2^31-1 -> E1 emits 0, state back to full;
    # finish(): carry_run=1, low < quarter -> bit 0 then one 1 (carry), padded: 0 0 1 00000 = 0x20
    e = czo.ArithmeticEncoder(); e.encode_counts(0, T // 2); assert e.finish() == bytes([0x20])
    # upper half: E2 emits 1; finish emits 0 then 1 -> bits 1 0 1 -> 0xA0
    e = czo.ArithmeticEncoder(); e.encode_counts(T // 2, T); assert e.finish() == bytes([0xA0])
    # empty stream: finish only: bits 0 1 -> 0x40
    e = czo.ArithmeticEncoder(); assert e.finish() == bytes([0x40])


def test_kat_middle_quarter_carry(czo):
    # [T/4, 3T/4): low=2^30, high=3*2^30-1 -> E3 (carry_run=1), low=0, high=2^32-1 after shift;
    # finish: carry_run=2, low<quarter -> 0 then two 1s -> 0 1 1 -> 0x60
    e = czo.ArithmeticEncoder(); e.encode_counts(T // 4, 3 * (T // 4)); assert e.finish() == bytes([0x60])


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_matches_python_restatement(czo, seed):
    rng = np.random.default_rng(seed)
    n = 40
    w = rng.integers(1, 1000, n); cdf = np.concatenate([[0], np.cumsum(w)]).astype(np.int64)
    cdf = (cdf * T // cdf[-1]).astype(np.uint32); cdf[-1] = T
    syms = rng.integers(0, n, 300)
    ce, pe = czo.ArithmeticEncoder(), PyEncoder()
    for s in syms:
        ce.encode_counts(int(cdf[s]), int(cdf[s + 1])); pe.encode_counts(int(cdf[s]), int(cdf[s + 1]), T)
    c, p = ce.finish(), pe.finish()
    assert c == p
    cd, pd = czo.ArithmeticDecoder(c), PyDecoder(c)
    for s in syms:
        assert cd.decode_symbol_counts(cdf) == pd.decode_symbol_counts(cdf.tolist(), T) == s


def test_zero_width_interval_is_reproduced_not_fixed(czo):
    # H3: c_lo == c_hi makes high = low - 1 (u128 wrap when 0); both restatements must agree bit for bit,
    # and the oracle must COUNT the event.
    cdf = np.array([0, 0, T // 2, T // 2, T], np.uint32)
    seq = [1, 3, 0, 1, 2, 3, 3, 1]
    ce, pe = czo.ArithmeticEncoder(), PyEncoder()
    for s in seq:
        ce.encode_counts(int(cdf[s]), int(cdf[s + 1])); pe.encode_counts(int(cdf[s]), int(cdf[s + 1]), T)
    c = ce.finish()
    assert c == pe.finish()
    assert ce.zero_width == 2          # symbols 0 and 2 are zero-width
    # the decoder can never return a zero-width symbol: decoding yields a different stream (the reference's latent bug)
    cd = czo.ArithmeticDecoder(c)
    dec = [cd.decode_symbol_counts(cdf) for _ in seq]
    assert dec != seq and all(d in (1, 3) for d in dec)


def test_decoder_pads_with_ones_past_eof(czo):
    d = czo.ArithmeticDecoder(b"")
    assert d.d.code == 0xFFFFFFFF
    cdf = np.array([0, T // 2, T], np.uint32)
    assert d.decode_symbol_counts(cdf) == 1


def test_softmax_pdf_matches_numpy_model(czo):
    rng = np.random.default_rng(0)
    lg = (rng.standard_normal(1000) * 4).astype(np.float32)
    pdf = czo.softmax_pdf(lg)
    # numpy model with the same scalar ops (f32 subtract, owned expf, sequential f64 sum)
    L = czo.lib()
    e = np.array([L.czo_expf(float(np.float32(x) - lg.max())) for x in lg], np.float64)
    s = 0.0
    for v in e:
        s += v
    assert np.array_equal(pdf, e / s)
    cdf = czo.quantize_pdf_to_cdf(pdf)
    acc, ref = 0.0, [0]
    for p in pdf:
        acc += p
        v = int(np.floor(acc * T)); v = min(max(v, 0), T); v = max(v, ref[-1]); ref.append(v)
    ref[-1] = T
    assert cdf.tolist() == ref and cdf[0] == 0 and cdf[-1] == T


def test_softmax_uniform_fallback_and_neg_inf(czo):
    lg = np.full(8, -np.inf, np.float32)        # max = -inf -> (l - max) = NaN -> expf NaN -> sum NaN: `sum <= 0` false
    pdf = czo.softmax_pdf(lg)
    assert np.isnan(pdf).all()
    lg = np.array([0, -np.inf, -200, -104.5], np.float32)
    pdf = czo.softmax_pdf(lg)
    assert pdf[0] == 1.0 and pdf[1] == 0.0 and pdf[2] == 0.0 and pdf[3] == 0.0


def test_rwkv_literals_cdf(czo):
    rng = np.random.default_rng(1)
    lg = (rng.standard_normal(512) * 6).astype(np.float32)
    pdf = czo.combined_pdf_with_literals(lg)
    assert pdf.size == 512 + 256 and abs(pdf.sum() - 1) < 1e-12
    assert np.all(pdf[512:] == pdf[512]) and pdf[512] > 0
    cdf = czo.logits_to_cdf(lg, "rwkv7")
    assert cdf.size == 512 + 257 and cdf[-1] == T and np.all(np.diff(cdf.astype(np.int64)) >= 1)   # floor => no zero width


@pytest.mark.parametrize("scale,expect_zero_width", [(1.0, False), (40.0, True)])
def test_logits_roundtrip_through_coder(czo, scale, expect_zero_width):
    rng = np.random.default_rng(7)
    V, n = 256, 400
    logits = (rng.standard_normal((n, V)) * scale).astype(np.float32)
    cdfs = [czo.logits_to_cdf(l) for l in logits]
    widths_zero = any((np.diff(c.astype(np.int64)) == 0).any() for c in cdfs)
    assert widths_zero == expect_zero_width
    # choose only encodable symbols (the reference silently corrupts otherwise; H3)
    syms = [int(rng.choice(np.nonzero(np.diff(c.astype(np.int64)) > 0)[0])) for c in cdfs]
    syms[0] = int(np.nonzero(np.diff(cdfs[0].astype(np.int64)) > 0)[0][0])
    e = czo.ArithmeticEncoder()
    for c, s in zip(cdfs, syms):
        e.encode_counts(int(c[s]), int(c[s + 1]))
    payload = e.finish()
    assert e.zero_width == 0
    d = czo.ArithmeticDecoder(payload)
    assert [d.decode_symbol_counts(c) for c in cdfs] == syms


def test_expf_properties(czo):
    L = czo.lib()
    assert L.czo_expf(0.0) == 1.0 and L.czo_expf(float("-inf")) == 0.0 and L.czo_expf(-104.0) == 0.0
    assert L.czo_expf(89.0) == float("inf") and np.isnan(L.czo_expf(float("nan")))
    xs = np.linspace(-87, 88, 5001).astype(np.float32)
    got = np.array([L.czo_expf(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 1.5e-7
    sub = L.czo_expf(-95.0)                       # subnormal result, rounded once
    assert 0 < sub < 1.2e-38 and abs(sub - np.exp(-95.0)) / np.exp(-95.0) < 1e-4
    assert np.all(np.diff(got) >= 0)

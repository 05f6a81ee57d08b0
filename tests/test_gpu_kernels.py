"""GPU parity tests proper (-m gpu): every HIP kernel, called through the C ABI / probe entry points of
libcandlezip_hip.so, against the CPU oracle on the same seeded inputs.  Bar: BIT-EXACT (u32 views compared) for
every f32 kernel, exact integers for the coder."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    m.lib()
    return m


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_bitexact(got, ref, what):
    g, r = _bits(got), _bits(ref)
    bad = np.nonzero(g != r)
    if bad[0].size:
        i = tuple(b[0] for b in bad)
        raise AssertionError(f"{what}: {bad[0].size}/{g.size} elements differ; first at {i}: got {got[i]!r} ref {ref[i]!r}")


@pytest.mark.parametrize("fn,name", [(0, "expf"), (1, "logf"), (2, "tanhf"), (5, "silu"), (6, "sigmoid"), (4, "sqrt")])
def test_scalar_functions_bitexact(cz, czo, fn, name):
    rng = np.random.default_rng(fn)
    x = np.concatenate([rng.standard_normal(200000) * 8, np.linspace(-110, 95, 50001), [0.0, -0.0, np.inf, -np.inf, 1e-40, -1e-40, 88.7, -87.4, -103.9]]).astype(np.float32)
    if name in ("logf", "sqrt"):
        x = np.abs(x)
    out = np.empty_like(x)
    assert cz.lib().cz_debug_scalar(fn, x.ctypes.data, None, out.ctypes.data, x.size) == 0
    L = czo.lib()
    ref_fn = {"expf": L.czo_expf, "logf": L.czo_logf, "tanhf": L.czo_tanhf}.get(name)
    if ref_fn:
        ref = np.array([ref_fn(float(v)) for v in x[:60000]], np.float32)
        _assert_bitexact(out[:60000], ref, name)
        ref2 = np.array([ref_fn(float(v)) for v in x[-9:]], np.float32)
        _assert_bitexact(out[-9:], ref2, name + " edge")
    elif name == "sqrt":
        _assert_bitexact(out, np.sqrt(x), name)                       # IEEE correctly rounded on both sides
    elif name == "silu":
        e = np.array([L.czo_expf(float(-v)) for v in x[:60000]], np.float32)
        _assert_bitexact(out[:60000], x[:60000] / (np.float32(1.0) + e), name)
    else:
        e = np.array([L.czo_expf(float(-v)) for v in x[:60000]], np.float32)
        _assert_bitexact(out[:60000], np.float32(1.0) / (e + np.float32(1.0)), name)


def test_f32_division_and_f64_ops_are_ieee(cz):
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(300000) * 10 ** rng.uniform(-20, 20, 300000)).astype(np.float32)
    b = (rng.standard_normal(300000) * 10 ** rng.uniform(-20, 20, 300000)).astype(np.float32)
    out = np.empty_like(a)
    assert cz.lib().cz_debug_scalar(3, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size) == 0
    with np.errstate(all="ignore"):
        _assert_bitexact(out, a / b, "f32 div")
    x = rng.random(300000) * 10 ** rng.uniform(-12, 0, 300000)
    y = rng.random(300000) * 4 + 1e-3
    o = np.empty_like(x)
    for fn, ref in ((0, x / y), (1, x + y), (2, np.floor(x * 1073741824.0)), (3, x * y)):
        assert cz.lib().cz_debug_f64(fn, x.ctypes.data, y.ctypes.data, o.ctypes.data, x.size) == 0
        assert np.array_equal(o.view(np.uint64), ref.view(np.uint64)), fn


def test_division_by_shared_reciprocal_is_exact(cz):
    """div_by() (csrc/cz_cdf_common.h): the CDF kernels' quotient by a row-wide divisor = one multiply + four fmas on the correctly rounded
    reciprocal; must equal the IEEE division the reference does (src/main.rs:799, 763-765, 779) bit for bit.  4 M operand pairs: f32-valued
    dividends (the exps), dividends that put the quotient next to a rounding midpoint, divisors with all-ones significands and just above 1."""
    rng = np.random.default_rng(20261004)
    n = 1 << 22
    b = np.where(rng.random(n) < 0.5, rng.uniform(1.0, 70000.0, n), np.floor(rng.uniform(1.0, 65793.0, n)))
    ones = rng.random(n) < 0.15
    b[ones] = np.ldexp(2.0 - (1 + rng.integers(0, 4, ones.sum())) * 2.0 ** -52, rng.integers(0, 17, ones.sum()))
    near1 = rng.random(n) < 0.1
    b[near1] = 1.0 + rng.random(near1.sum()) * 2.0 ** -rng.integers(1, 40, near1.sum()).astype(np.float64)
    a = np.ldexp(rng.random(n), -rng.integers(0, 140, n)).astype(np.float32).astype(np.float64)
    mid = rng.random(n) < 0.5
    q = np.ldexp(1.0 + rng.random(mid.sum()), -rng.integers(0, 60, mid.sum()))
    am = b[mid] * q                                                     # then walk a few ulps around: some of these quotients sit at midpoints
    am = am.view(np.int64) + rng.integers(-2, 3, mid.sum())
    a[mid] = am.view(np.float64)
    a[:6] = [0.0, 1.0, 2.0 ** -149, 1.862645149230957e-09, b[4], np.nextafter(b[5], 0.0)]
    o = np.empty_like(a)
    assert cz.lib().cz_debug_f64(4, a.ctypes.data, b.ctypes.data, o.ctypes.data, n) == 0
    ref = a / b
    bad = np.flatnonzero(o.view(np.uint64) != ref.view(np.uint64))
    assert bad.size == 0, (bad[:5], a[bad[:5]], b[bad[:5]])


@pytest.mark.parametrize("M,N,K,big", [(300, 576, 576, 1), (130, 128, 1536, 1), (1000, 960, 576, 1), (257, 3072, 576, 1), (64, 576, 1536, 1),
                                       (1, 576, 576, 0), (7, 960, 576, 0), (48, 576, 1536, 0), (64, 3072, 576, 0), (33, 512, 128, 0), (5, 49152, 576, 0)])
def test_gemm_matches_canonical_fmaf_chain(cz, czo, M, N, K, big):
    """the MFMA chain must equal the oracle's k-ascending fmaf chain per 64-chunk, bit for bit (DESIGN.md 'Numerics')"""
    rng = np.random.default_rng(M * 7 + N)
    A = (rng.standard_normal((M, K)) * rng.choice([1e-3, 1.0, 30.0], (M, 1))).astype(np.float32)
    W = czo.to_bf16(rng.standard_normal((N, K)) * 0.05)
    R = rng.standard_normal((M, N)).astype(np.float32)
    for res in (None, R):
        C = np.empty((M, N), np.float32)
        rc = cz.lib().cz_debug_gemm(A.ctypes.data, W.ctypes.data, res.ctypes.data if res is not None else None, C.ctypes.data, M, N, K, big)
        assert rc == 0, cz.lib().cz_last_error()
        _assert_bitexact(C, czo.gemm_bf16(A, W, res), f"gemm {M}x{N}x{K} big={big} res={res is not None}")


def test_gemm_big_and_skinny_kernels_agree(cz, czo):
    rng = np.random.default_rng(5)
    M, N, K = 48, 960, 576
    A = rng.standard_normal((M, K)).astype(np.float32); W = czo.to_bf16(rng.standard_normal((N, K)) * 0.05)
    C0 = np.empty((M, N), np.float32); C1 = np.empty((M, N), np.float32)
    assert cz.lib().cz_debug_gemm(A.ctypes.data, W.ctypes.data, None, C0.ctypes.data, M, N, K, 0) == 0
    assert cz.lib().cz_debug_gemm(A.ctypes.data, W.ctypes.data, None, C1.ctypes.data, M, N, K, 1) == 0
    _assert_bitexact(C0, C1, "skinny vs big")


@pytest.mark.parametrize("M,d", [(1, 576), (37, 576), (100, 128), (16, 1024)])
def test_rmsnorm_bitexact(cz, czo, M, d):
    rng = np.random.default_rng(M)
    X = (rng.standard_normal((M, d)) * 3).astype(np.float32); w = czo.to_bf16(1 + rng.standard_normal(d) * 0.1)
    Y = np.empty_like(X)
    assert cz.lib().cz_debug_rmsnorm(X.ctypes.data, w.ctypes.data, Y.ctypes.data, M, d, 1e-5) == 0
    _assert_bitexact(Y, czo.rmsnorm_bf16w(X, w, 1e-5), "rmsnorm")


@pytest.mark.parametrize("M,pos0,nh,nkv,variant", [
    (1, 0, 9, 3, 0), (1, 700, 9, 3, 0), (40, 0, 9, 3, 0), (130, 0, 2, 1, 0), (3, 1000, 9, 3, 0), (70, 64, 4, 4, 0), (17, 63, 2, 1, 0),
    # BASELINE config 5's range (keys up to 1535 / 2048): one row at the sawtooth's peak, the register-resident kernel at its last chunk, v1's last keys
    (1, 1535, 9, 3, 0), (3, 2040, 9, 3, 0), (511, 0, 9, 3, 0), (40, 1490, 9, 3, 0), (20, 2020, 2, 1, 0),
    # the three-pass kernel (attention_v3_kernel): forced on shapes the others cover too, then past 2048 keys where it is the only one
    (40, 0, 9, 3, 2), (70, 1500, 9, 3, 2), (33, 100, 4, 4, 2), (1, 300, 2, 1, 2), (5, 3000, 9, 3, 0), (300, 2500, 2, 1, 0), (1, 8191, 9, 3, 0), (17, 4090, 4, 1, 0)])
def test_attention_bitexact(cz, czo, M, pos0, nh, nkv, variant):
    rng = np.random.default_rng(M + pos0)
    nk = pos0 + M
    Q = rng.standard_normal((M, nh * 64)).astype(np.float32)
    K = rng.standard_normal((nk, nkv * 64)).astype(np.float32); V = rng.standard_normal((nk, nkv * 64)).astype(np.float32)
    O = np.empty_like(Q)
    cz.lib().cz_debug_attn_variant(variant)
    try:
        rc = cz.lib().cz_debug_attention(Q.ctypes.data, K.ctypes.data, V.ctypes.data, O.ctypes.data, M, pos0, nh, nkv)
    finally:
        cz.lib().cz_debug_attn_variant(0)
    assert rc == 0, cz.lib().cz_last_error()
    _assert_bitexact(O, czo.attention_ref(Q, K, V, pos0, nh, nkv), "attention")


def _cdf_pairs(cz, logits, sym, V, mode, serial):
    rows = logits.shape[0]
    lo = np.empty(rows, np.uint32); hi = np.empty(rows, np.uint32); fb = np.zeros(1, np.uint32)
    rc = cz.lib().cz_debug_cdf_pairs(logits.ctypes.data, sym.ctypes.data, lo.ctypes.data, hi.ctypes.data, rows, V, mode, serial, fb.ctypes.data)
    assert rc == 0, cz.lib().cz_last_error()
    return lo, hi, int(fb[0])


@pytest.mark.parametrize("serial", [1, 0])
@pytest.mark.parametrize("V,mode,scale", [(512, 0, 1.0), (512, 0, 30.0), (49152, 0, 2.0), (49152, 0, 12.0), (512, 1, 3.0), (65536, 1, 4.0)])
def test_cdf_pairs_exact(cz, czo, V, mode, scale, serial):
    """serial = the reference's own walk, one lane per row; serial=0 = the exact-parallel kernel (mode 0) — both must equal the oracle"""
    rng = np.random.default_rng(V + mode)
    rows = 70 if V < 1000 else 9
    logits = (rng.standard_normal((rows, V)) * scale).astype(np.float32)
    nsym = V if mode == 0 else V + 256
    sym = rng.integers(0, nsym, rows).astype(np.uint32)
    sym[0], sym[1] = 0, nsym - 1
    lo, hi, _ = _cdf_pairs(cz, logits, sym, V, mode, serial)
    for r in range(rows):
        cdf = czo.logits_to_cdf(logits[r], "smollm" if mode == 0 else "rwkv7")
        assert (lo[r], hi[r]) == (cdf[sym[r]], cdf[sym[r] + 1]), (r, sym[r])


def _adversarial_rows(V, rng):
    """rows built to stress the exact-parallel chain: binade crossings everywhere, ties, zeros, one-hot, flat"""
    rows = []
    rows.append(np.sort(rng.standard_normal(V) * 25).astype(np.float32))                   # ascending: crossings at almost every step early
    rows.append(np.sort(rng.standard_normal(V) * 25)[::-1].astype(np.float32).copy())      # descending: max first, then ever smaller terms
    rows.append((np.round(rng.standard_normal(V) * 20) * np.float32(0.6931472)).astype(np.float32))   # exps ~ powers of two: ties
    rows.append(np.zeros(V, np.float32))                                                   # flat: p = 1/V exactly representable? no -> rounding each step
    onehot = np.full(V, -np.inf, np.float32); onehot[V // 3] = 5.0
    rows.append(onehot)                                                                    # exp(-inf) = 0 everywhere else
    r = (rng.standard_normal(V) * 3).astype(np.float32); r[: V // 2] -= 120.0               # first half underflows to subnormal/zero exps
    rows.append(r)
    r = (rng.standard_normal(V) * 3).astype(np.float32); r[0] = 60.0                        # huge first term: acc >= 1 from the start
    rows.append(r)
    r = (rng.standard_normal(V) * 3).astype(np.float32); r[-1] = 60.0                       # huge last term
    rows.append(r)
    for s in (0.5, 4.0, 15.0, 40.0):
        rows.append((rng.standard_normal(V) * s).astype(np.float32))
    return np.stack(rows)


@pytest.mark.parametrize("V", [1, 2, 511, 1000, 1024, 1025, 40000, 49152])
def test_exact_parallel_cdf_on_adversarial_rows(cz, czo, V):
    rng = np.random.default_rng(V)
    logits = _adversarial_rows(V, rng)
    rows = logits.shape[0]
    fallbacks = 0
    for trial in range(3):
        sym = rng.integers(0, V, rows).astype(np.uint32)
        if trial == 0:
            sym[:] = V - 1
        if trial == 1:
            sym[:] = 0
        lo, hi, fb = _cdf_pairs(cz, logits, sym, V, 0, 0)
        fallbacks += fb
        for r in range(rows):
            cdf = czo.logits_to_cdf(logits[r])
            assert (lo[r], hi[r]) == (cdf[sym[r]], cdf[sym[r] + 1]), (V, trial, r, sym[r])
    # the uniform/NaN rows legitimately take the serial walk; everything else must have stayed on the parallel path
    assert fallbacks <= 3 * 2, fallbacks


@pytest.mark.parametrize("V", [1, 2, 300, 769, 1024, 20000, 65536])
def test_exact_parallel_cdf_literals_mode_on_adversarial_rows(cz, czo, V):
    """RWKV mode (floor, renormalise, mix in 256 literal entries, renormalise, running cdf: four sequential f64 chains,
    src/main.rs:758-782, 2303-2325) through the exact-parallel kernel, on the same adversarial rows; the symbol may be a literal."""
    rng = np.random.default_rng(V + 77)
    logits = _adversarial_rows(V, rng)
    rows = logits.shape[0]
    n = V + 256
    fallbacks = 0
    for trial in range(4):
        sym = rng.integers(0, n, rows).astype(np.uint32)
        if trial == 0:
            sym[:] = n - 1
        if trial == 1:
            sym[:] = 0
        if trial == 2:
            sym[:] = rng.integers(max(V - 2, 0), V + 3, rows)      # around the vocabulary / literal boundary
        lo, hi, fb = _cdf_pairs(cz, logits, sym, V, 1, 0)
        fallbacks += fb
        for r in range(rows):
            cdf = czo.logits_to_cdf(logits[r], "rwkv7")
            assert (lo[r], hi[r]) == (cdf[sym[r]], cdf[sym[r] + 1]), (V, trial, r, sym[r])
    assert fallbacks == 0, fallbacks            # the floor keeps every chain strictly positive: nothing may leave the parallel path


@pytest.mark.parametrize("serial", [1, 0])
@pytest.mark.parametrize("V,mode,scale", [(512, 0, 2.0), (512, 1, 2.0), (512, 0, 40.0), (3000, 0, 6.0), (3000, 1, 6.0)])
def test_device_arithmetic_decoder_matches_oracle(cz, czo, V, mode, scale, serial):
    rng = np.random.default_rng(V * 3 + mode)
    steps = 300
    logits = (rng.standard_normal((steps, V)) * scale).astype(np.float32)
    cdfs = [czo.logits_to_cdf(l, "smollm" if mode == 0 else "rwkv7") for l in logits]
    syms = [int(rng.choice(np.nonzero(np.diff(c.astype(np.int64)) > 0)[0])) for c in cdfs]
    e = czo.ArithmeticEncoder()
    for c, s in zip(cdfs, syms):
        e.encode_counts(int(c[s]), int(c[s + 1]))
    payload = e.finish()
    buf = np.frombuffer(payload, np.uint8).copy()
    out = np.zeros(steps, np.uint32)
    rc = cz.lib().cz_debug_decode_symbols(logits.ctypes.data, steps, V, mode, buf.ctypes.data, len(payload), out.ctypes.data, serial)
    assert rc == 0, cz.lib().cz_last_error()
    assert out.tolist() == syms
    # a corrupt / truncated payload must decode to exactly what the reference decoder yields (EOF bits read as 1)
    cut = payload[: len(payload) // 2]
    d = czo.ArithmeticDecoder(cut)
    ref = [d.decode_symbol_counts(c) for c in cdfs]
    buf2 = np.frombuffer(cut, np.uint8).copy()
    rc = cz.lib().cz_debug_decode_symbols(logits.ctypes.data, steps, V, mode, buf2.ctypes.data, len(cut), out.ctypes.data, serial)
    assert rc == 0 and out.tolist() == ref


def test_decoder_on_49152_vocab_parallel_equals_serial(cz, czo):
    rng = np.random.default_rng(99)
    V, steps = 49152, 40
    logits = (rng.standard_normal((steps, V)) * 5).astype(np.float32)
    cdfs = [czo.logits_to_cdf(l) for l in logits]
    syms = [int(rng.choice(np.nonzero(np.diff(c.astype(np.int64)) > 0)[0])) for c in cdfs]
    e = czo.ArithmeticEncoder()
    for c, s in zip(cdfs, syms):
        e.encode_counts(int(c[s]), int(c[s + 1]))
    payload = e.finish()
    buf = np.frombuffer(payload, np.uint8).copy()
    for serial in (0, 1):
        out = np.zeros(steps, np.uint32)
        rc = cz.lib().cz_debug_decode_symbols(logits.ctypes.data, steps, V, 0, buf.ctypes.data, len(payload), out.ctypes.data, serial)
        assert rc == 0 and out.tolist() == syms, serial

"""CPU-side checks of the product boundary: the C-ABI library loads and exports every symbol include/candlezip.h
declares; host-only entry points (coder, container, BLAKE3) agree with the oracle.  No GPU compute is called."""
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cz():
    import candlezip_amd as m
    if not os.path.exists(m.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    m.lib()
    return m


def test_library_exports_every_declared_symbol(cz):
    hdr = open(os.path.join(ROOT, "include", "candlezip.h")).read()
    declared = set(re.findall(r"\b(cz_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(cz.EXPORTS), declared ^ set(cz.EXPORTS)
    L = cz.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_no_gpu_means_loud_failure_not_fallback(cz):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cz.CandlezipError, match="HIP"):
        cz.Session("smollm", cz.tiny_llama_config())


def test_host_coder_matches_oracle(cz, czo):
    rng = np.random.default_rng(0)
    T = 1 << 30
    cuts = np.sort(rng.integers(0, T, (500, 2)), axis=1).astype(np.uint32)
    cuts[7] = (5, 5); cuts[9] = (0, 0)                           # zero-width events incl. the u128 wrap case
    lo, hi = cuts[:, 0].copy(), cuts[:, 1].copy()
    pay, zw = cz.ac_encode_pairs(lo, hi)
    ref, zw_ref = czo.encode_pairs(lo, hi)
    assert pay == ref and zw == zw_ref >= 2
    assert cz.ac_encode_pairs(np.zeros(0, np.uint32), np.zeros(0, np.uint32))[0] == bytes([0x40])


def test_container_and_blake3_match_reference_files(cz, czo, golden):
    for name, inp in (("asyoulik", "asyoulik.txt"), ("fields", "fields.c"), ("enwik8_128kb_0", "enwik8_128kb_0")):
        buf = open(os.path.join(golden, "ref_canz", name + ".canz"), "rb").read()
        h, repr_, n = cz.read_header(buf)
        ho, ro, no = czo.read_header(buf)
        assert (h.token_count, h.orig_len_bytes, h.vocab_size, h.context_window, h.reprime_interval, h.reserved_flags, n) == \
               (ho.token_count, ho.orig_len_bytes, ho.vocab_size, ho.context_window, ho.reprime_interval, ho.reserved_flags, no)
        assert cz.write_header(h, repr_) == buf[:n]
        data = open(os.path.join(golden, "inputs", inp), "rb").read()
        assert cz.blake3_16(data) == bytes(h.orig_hash16) == czo.blake3_16(data)
    with pytest.raises(cz.CandlezipError, match="bad magic"):
        cz.read_header(b"XXXX" + bytes(120))
    with pytest.raises(cz.CandlezipError, match="bad version"):
        cz.read_header(b"GTPZ\x03\x00" + bytes(120))
    assert cz.lib().cz_flags_pack(0, 0, 0, 512) == 0x02000000


def test_gate_section_skipper(cz, golden):
    import ctypes as C
    buf = open(os.path.join(golden, "ref_canz", "alice29.canz"), "rb").read()
    h, _, n = cz.read_header(buf)
    assert h.reserved_flags & 4
    a = np.frombuffer(buf[n:], np.uint8)
    used = C.c_size_t()
    assert cz.lib().cz_skip_gate_section(a.ctypes.data, a.size, C.byref(used)) == 0 and used.value == 5   # "AGT2" + varint(0)
    bad = np.frombuffer(b"NOPE" + bytes(8), np.uint8)
    assert cz.lib().cz_skip_gate_section(bad.ctypes.data, bad.size, C.byref(used)) == 6


def test_product_does_not_touch_the_oracle():
    """a product path that routes through oracle/ voids every parity claim: nothing under candlezip_amd/ or include/ may
    mention it"""
    for base in ("candlezip_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    assert "czo_" not in txt and "cz_oracle" not in txt and "oracle/" not in txt.replace("CPU oracle", ""), os.path.join(dp, f)


def test_incremental_encoder_equals_one_shot(cz):
    """cz_ac_encoder_* (ArithmeticEncoder::encode_counts called window by window): the pieces concatenate to cz_ac_encode_pairs' payload"""
    import numpy as np
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 3000):
        lo = np.sort(rng.integers(0, 1 << 30, (n, 2)), axis=1).astype(np.uint32)
        lo[:, 1] = np.maximum(lo[:, 1], lo[:, 0] + (rng.integers(0, 4, n) > 0))        # a few zero-width intervals too
        ref, zw = cz.ac_encode_pairs(lo[:, 0], lo[:, 1])
        e = cz.AcEncoder()
        got = b""; k = 0
        for w in (1, 5, 0, 250, 10 ** 6):
            got += e.push(lo[k:k + w, 0], lo[k:k + w, 1]); k += w
            assert ref.startswith(got)                                                  # every piece is final when it comes out
        tail, zw2 = e.finish()
        assert got + tail == ref and zw2 == zw
        with pytest.raises(cz.CandlezipError):
            e.push(lo[:1, 0], lo[:1, 1])
        e.close()

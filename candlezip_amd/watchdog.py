"""Watchdog-lite: the self-test diagnostics of the reference (src/main.rs:160-171, 195-225, 2629-2647) without the agent.

The reference records one JSON line per coded step on both sides and, while decoding, reports the FIRST step whose decoded
symbol differs from the encoded one (`watchdog_mismatch.json`: phase, i, encode_sym, decode_sym, note).  Here both id sequences
exist as arrays after the lock-step decode, so the report is one vectorised comparison.  The per-step logits digest of the
reference (`pdf_digest`, only with CANDLEZIP_WATCHDOG_DIGEST) is not produced: logits never leave HBM on this path; the field is
written as null, exactly as the reference does with digests off.
"""
import json
import os
import time

import numpy as np


def enabled():
    return os.environ.get("CANDLEZIP_WATCHDOG", "") not in ("", "0")                    # src/main.rs:171


def run_dir(input_path, backend):
    """scan_output/<stem>_selftest_<backend>_<UTC ts> (src/main.rs:160-169), or CANDLEZIP_WATCHDOG_DIR when set"""
    d = os.environ.get("CANDLEZIP_WATCHDOG_DIR")
    if not d:
        stem = os.path.splitext(os.path.basename(input_path))[0] or "input"
        d = os.path.join("scan_output", f"{stem}_selftest_{backend.lower()}_{time.strftime('%Y%m%d_%H%M%S', time.gmtime())}")
    os.makedirs(d, exist_ok=True)
    return d


def first_mismatch(encode_ids, decode_ids):
    """(i, encode_sym, decode_sym) of the first differing coded symbol (ids exclude BOS), a length difference counts as a
    mismatch at the shorter length (sym = None on the side that ended); None when identical"""
    a = np.asarray(encode_ids, np.int64); b = np.asarray(decode_ids, np.int64)
    n = min(a.size, b.size)
    diff = np.nonzero(a[:n] != b[:n])[0]
    if diff.size:
        i = int(diff[0])
        return i, int(a[i]), int(b[i])
    if a.size != b.size:
        return n, (int(a[n]) if a.size > n else None), (int(b[n]) if b.size > n else None)
    return None


def zero_width_steps(c_lo, c_hi, limit=16):
    """indices of coded steps whose interval is empty (c_lo == c_hi): the reference encodes them, and its decoder diverges there
    (SURVEY H3); the first of them is where a mismatch report will point"""
    z = np.nonzero(np.asarray(c_lo) == np.asarray(c_hi))[0]
    return [int(i) for i in z[:limit]], int(z.size)


def write_steps(dirpath, name, ids, phase):
    """watchdog_{encode,decode}_steps.jsonl: one record per coded step, then the `*_done` record (src/main.rs:2641-2647)"""
    with open(os.path.join(dirpath, name), "w") as f:
        for i, s in enumerate(np.asarray(ids).tolist()):
            f.write(json.dumps({"phase": f"{phase}_step", "i": i, "sym": int(s), "pdf_digest": None}) + "\n")
        f.write(json.dumps({"phase": f"{phase}_done", "tokens": len(ids) + 1}) + "\n")


def report(dirpath, encode_ids, decode_ids, c_lo=None, c_hi=None):
    """writes watchdog_mismatch.json when the sequences differ; returns the record (or None)"""
    mm = first_mismatch(encode_ids, decode_ids)
    if mm is None:
        return None
    rec = {"phase": "mismatch", "i": mm[0], "encode_sym": mm[1], "decode_sym": mm[2], "note": "first differing symbol"}
    if c_lo is not None and c_hi is not None:
        idx, n = zero_width_steps(c_lo, c_hi)
        rec["zero_width_steps"] = idx
        rec["zero_width_total"] = n
    if dirpath:
        with open(os.path.join(dirpath, "watchdog_mismatch.json"), "w") as f:
            json.dump(rec, f)
    return rec

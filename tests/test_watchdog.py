"""Watchdog-lite: first-mismatch report and step traces, mirroring src/main.rs:2629-2647 — host logic only."""
import json
import os
import numpy as np


def test_first_mismatch_and_report(tmp_path, monkeypatch):
    from candlezip_amd import watchdog as w
    enc = np.array([5, 6, 7, 8, 9], np.uint32)
    assert w.first_mismatch(enc, enc.copy()) is None
    assert w.first_mismatch(enc, [5, 6, 1, 8, 2]) == (2, 7, 1)
    assert w.first_mismatch(enc, enc[:3]) == (3, 8, None)                 # decoder stopped early
    assert w.first_mismatch(enc[:2], enc) == (2, None, 7)
    assert w.first_mismatch([], []) is None
    lo = np.array([0, 10, 20, 20, 30], np.uint32); hi = np.array([10, 20, 20, 30, 30], np.uint32)
    assert w.zero_width_steps(lo, hi) == ([2, 4], 2)
    rec = w.report(str(tmp_path), enc, [5, 6, 1, 8, 2], lo, hi)
    on_disk = json.load(open(tmp_path / "watchdog_mismatch.json"))
    assert rec == on_disk and on_disk["phase"] == "mismatch" and on_disk["i"] == 2 and on_disk["encode_sym"] == 7 and on_disk["decode_sym"] == 1
    assert on_disk["note"] == "first differing symbol" and on_disk["zero_width_steps"] == [2, 4]
    assert w.report(str(tmp_path / "none"), enc, enc) is None
    # step traces: one record per coded step, then the done record carrying tokens = steps + BOS
    w.write_steps(str(tmp_path), "watchdog_decode_steps.jsonl", enc, "decode")
    lines = [json.loads(l) for l in open(tmp_path / "watchdog_decode_steps.jsonl")]
    assert lines[0] == {"phase": "decode_step", "i": 0, "sym": 5, "pdf_digest": None} and lines[-1] == {"phase": "decode_done", "tokens": 6}
    # directory naming and the env switches of the reference
    monkeypatch.delenv("CANDLEZIP_WATCHDOG", raising=False)
    assert not w.enabled()
    monkeypatch.setenv("CANDLEZIP_WATCHDOG", "1")
    assert w.enabled()
    monkeypatch.setenv("CANDLEZIP_WATCHDOG_DIR", str(tmp_path / "wd"))
    assert w.run_dir("x/alice29.txt", "smollm") == str(tmp_path / "wd") and os.path.isdir(tmp_path / "wd")
    monkeypatch.delenv("CANDLEZIP_WATCHDOG_DIR")
    monkeypatch.chdir(tmp_path)
    d = w.run_dir("x/alice29.txt", "SmolLM")
    assert d.startswith(os.path.join("scan_output", "alice29_selftest_smollm_")) and os.path.isdir(d)

"""CLI surface (src/main.rs:34-141): flag names/defaults are kept, agent flags are rejected loudly, no silent CPU path."""
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    return subprocess.run([sys.executable, "-m", "candlezip_amd", *argv], cwd=ROOT, capture_output=True, text=True)


def test_defaults_match_reference_cli():
    sys.path.insert(0, ROOT)
    from candlezip_amd.cli import build_parser
    a = build_parser().parse_args(["compress", "in", "out"])
    assert (a.backend, a.context, a.reprime_interval, a.cpu, a.force_mismatch) == ("smollm", 512, 512, False, False)
    assert a.model_repo == "HuggingFaceTB/SmolLM2-135M" and a.tokenizer_file == "tokenizer.json"
    assert a.rwkv_model == "modeldir/prepared/model.safetensors" and a.rwkv_tokenizer == "modeldir/rwkv_vocab_v20230424.json"
    b = build_parser().parse_args(["--backend", "rwkv7", "--context", "1024", "--reprime-interval", "1024", "self-test", "f"])
    assert (b.backend, b.context, b.reprime_interval, b.command, b.input) == ("rwkv7", 1024, 1024, "self-test", "f")


def test_agent_flags_and_cpu_are_rejected_not_ignored():
    r = _run("--agent", "compress", "a", "b")
    assert r.returncode != 0 and "not supported in this build" in r.stderr
    r = _run("--scan-lookahead", "64", "self-test", "a")
    assert r.returncode != 0 and "not supported" in r.stderr
    r = _run("--cpu", "self-test", "a")
    assert r.returncode != 0 and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_self_test_roundtrip_rwkv7_literals_and_smollm_ids(tmp_path):
    # rwkv7 without a vocab file: every byte is a literal symbol (src/main.rs:863-868) -> exact byte round trip
    src = tmp_path / "sample.bin"
    src.write_bytes(bytes(range(256)) + b"hello candlezip")
    r = _run("--backend", "rwkv7", "--synthetic", "7", "--rwkv-tokenizer", str(tmp_path / "missing.json"), "self-test", str(src))
    assert r.returncode == 0 and "Roundtrip OK" in r.stdout, r.stderr + r.stdout
    assert "Bits per byte" in r.stdout and "Compression" in r.stdout
    # smollm with token ids: compress, then decompress back to ids
    ids = np.random.default_rng(0).integers(0, 49152, 700).astype("<u4")
    idf = tmp_path / "tok.ids"; ids.tofile(idf)
    inp = tmp_path / "in.txt"; inp.write_bytes(b"x" * 1000)
    out = tmp_path / "o.canz"; back = tmp_path / "back.ids"
    r = _run("--synthetic", "1", "--token-ids", str(idf), "compress", str(inp), str(out))
    assert r.returncode == 0 and "bits/byte=" in r.stdout, r.stderr
    r = _run("--synthetic", "1", "--token-ids", str(idf), "decompress", str(out), str(back))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(back, "<u4"), ids)
    # a different model must be refused unless --force-mismatch (src/main.rs:2450-2459)
    r = _run("--synthetic", "2", "--token-ids", str(idf), "decompress", str(out), str(back))
    assert r.returncode != 0 and "model hash mismatch" in r.stderr


def test_context_window_checks():
    """context 0 is refused like the reference's empty-history error; context < 512 desynchronises encoder and decoder in the reference
    (src/main.rs:1935 vs :2506) and is announced loudly"""
    sys.path.insert(0, ROOT)
    from candlezip_amd import cli
    import io, contextlib
    a = cli.build_parser().parse_args(["--context", "0", "compress", "a", "b"])
    with pytest.raises(SystemExit, match="context must be > 0"):
        cli._check_window(a)
    a = cli.build_parser().parse_args(["--context", "300", "compress", "a", "b"])
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        cli._check_window(a)
    assert "will NOT decode" in err.getvalue()
    a = cli.build_parser().parse_args(["--backend", "rwkv7", "--context", "0", "compress", "a", "b"])
    cli._check_window(a)                                                            # RWKV7 never re-primes: the window is unused


@pytest.mark.gpu
def test_producer_mark_guards_foreign_archives(tmp_path):
    """a .canz without this build's producer mark (reserved_flags bit 15) — e.g. one written by the candle reference for the same weights — is
    refused unless --force-mismatch: its payload only decodes with the arithmetic that wrote it"""
    ids = np.random.default_rng(1).integers(0, 49152, 40).astype("<u4")
    idf = tmp_path / "tok.ids"; ids.tofile(idf)
    inp = tmp_path / "in.txt"; inp.write_bytes(b"x" * 100)
    out = tmp_path / "o.canz"; back = tmp_path / "back.ids"
    r = _run("--synthetic", "1", "--token-ids", str(idf), "compress", str(inp), str(out))
    assert r.returncode == 0, r.stderr
    blob = bytearray(out.read_bytes())
    import candlezip_amd as cz
    h, _, _ = cz.read_header(bytes(blob))
    assert h.reserved_flags == (512 << 16) | cz.FLAG_PRODUCER_HIP
    k = bytes(blob).index(int(h.reserved_flags).to_bytes(4, "little"))
    blob[k:k + 4] = (512 << 16).to_bytes(4, "little")                              # what the reference writes for this file (flags_pack, :247-255)
    foreign = tmp_path / "foreign.canz"; foreign.write_bytes(bytes(blob))
    r = _run("--synthetic", "1", "--token-ids", str(idf), "decompress", str(foreign), str(back))
    assert r.returncode != 0 and "producer mismatch" in r.stderr
    r = _run("--synthetic", "1", "--token-ids", str(idf), "--force-mismatch", "decompress", str(foreign), str(back))
    assert r.returncode == 0 and "producer mark" in r.stderr and np.array_equal(np.fromfile(back, "<u4"), ids)


@pytest.mark.gpu
def test_self_test_rwkv7_with_world_vocabulary(tmp_path):
    golden = os.path.join(ROOT, "tests", "golden")
    src = tmp_path / "fields.c"
    src.write_bytes(open(os.path.join(golden, "inputs", "fields.c"), "rb").read()[:6000])
    r = _run("--backend", "rwkv7", "--synthetic", "7", "--rwkv-tokenizer", os.path.join(golden, "world_vocab_test.json"), "self-test", str(src))
    assert r.returncode == 0 and "Roundtrip OK" in r.stdout, r.stderr + r.stdout


@pytest.mark.gpu
def test_self_test_text_roundtrip_with_in_process_bpe(tmp_path):
    """the whole drop-in path on text: tokenizer.json (in-process byte-level BPE) -> ids -> GPU encode -> .canz -> GPU decode -> text"""
    golden = os.path.join(ROOT, "tests", "golden")
    src = tmp_path / "excerpt.txt"
    src.write_bytes(open(os.path.join(golden, "inputs", "alice29.txt"), "rb").read()[2000:5000] + "\n caf\u00e9 1234 \u4e2d\u6587\n".encode())
    r = _run("--synthetic", "3", "--tokenizer", os.path.join(golden, "bpe_test_tokenizer.json"), "self-test", str(src))
    assert r.returncode == 0 and "Roundtrip OK" in r.stdout, r.stderr + r.stdout
    assert (tmp_path / "excerpt.canz").read_bytes()[:4] == b"GTPZ" and (tmp_path / "excerpt.roundtrip.txt").read_bytes() == src.read_bytes()


@pytest.mark.gpu
def test_corpus_self_test_on_reference_inputs():
    """BASELINE config 3 on the committed reference inputs: 64 KiB chunks of alice29 / asyoulik / enwik8_128kb_0 / fields.c, tokenised
    in-process, compressed and decompressed as one lock-step batch; every chunk must come back (up to the reference's own from_utf8_lossy)"""
    import json
    golden = os.path.join(ROOT, "tests", "golden")
    files = [os.path.join(golden, "inputs", n) for n in ("alice29.txt", "asyoulik.txt", "enwik8_128kb_0", "fields.c")]
    r = subprocess.run([sys.executable, "-m", "candlezip_amd.corpus", "--synthetic", "5", "--tokenizer", os.path.join(golden, "bpe_test_tokenizer.json"),
                        "--chunk-bytes", "16384", "--max-streams", "32", *files], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["roundtrip_failures"] == [] and d["chunks"] == 10 + 8 + 8 + 1 and d["tokens"] > 100000 and d["self_test_tokens_per_s"] > 0


@pytest.mark.gpu
def test_reprime_interval_beyond_2048_positions_and_the_model_limit(tmp_path):
    """`--reprime-interval` is bounded by the model only (src/main.rs:1935, 2280-2284, 2506): 2 600 tokens at interval 2 000 reach 2 511 positions
    (SmolLM2-135M shape, max_position_embeddings 8192) and round-trip through the CLI; an interval whose schedule would pass 8192 positions is
    refused before anything runs, where the reference would fail in the middle of the file"""
    ids = np.random.default_rng(3).integers(0, 49152, 2600).astype("<u4")
    idf = tmp_path / "tok.ids"; ids.tofile(idf)
    inp = tmp_path / "in.txt"; inp.write_bytes(b"z" * 5000)
    out = tmp_path / "o.canz"; back = tmp_path / "back.ids"
    r = _run("--synthetic", "1", "--token-ids", str(idf), "--reprime-interval", "2000", "compress", str(inp), str(out))
    assert r.returncode == 0 and "bits/byte=" in r.stdout, r.stderr
    r = _run("--synthetic", "1", "--token-ids", str(idf), "decompress", str(out), str(back))
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(back, "<u4"), ids)
    big = np.random.default_rng(4).integers(0, 49152, 16500).astype("<u4")
    bigf = tmp_path / "big.ids"; big.tofile(bigf)
    r = _run("--synthetic", "1", "--token-ids", str(bigf), "--reprime-interval", "8000", "compress", str(inp), str(tmp_path / "no.canz"))
    assert r.returncode != 0 and "max_position_embeddings" in r.stderr, r.stderr[-500:]

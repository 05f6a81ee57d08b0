"""SURVEY §8e on one GPU box: two ranks (one process each, gloo rendezvous on 127.0.0.1, both on GPU 0 — the driver runs the RCCL / 8-GPU
case) must produce exactly what one rank produces: (a) the corpus driver shards chunks over ranks (LPT, no data-path collective);
(b) ONE file's encode is sharded by re-prime segments, the (c_lo, c_hi) pairs are gathered on rank 0, which writes the single .canz."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ranks(world, argv, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CZ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, *argv], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:] + out[-1000:]
        outs.append(out)
    return outs


def test_corpus_two_ranks_equal_one_rank():
    golden = os.path.join(ROOT, "tests", "golden")
    files = [os.path.join(golden, "inputs", n) for n in ("fields.c", "alice29.txt")]
    argv = ["-m", "candlezip_amd.corpus", "--synthetic", "5", "--tokenizer", os.path.join(golden, "bpe_test_tokenizer.json"), "--chunk-bytes", "16384",
            "--max-streams", "16", *files]
    one = json.loads(subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    two = json.loads(_ranks(2, argv, 29611 + os.getpid() % 500)[0].strip().splitlines()[-1])
    assert one["roundtrip_failures"] == [] == two["roundtrip_failures"] and two["ranks"] == 2
    for k in ("chunks", "tokens", "payload_bytes", "payload_digest"):
        assert one[k] == two[k], k


def test_single_file_encode_sharded_over_two_ranks_writes_the_same_canz(tmp_path):
    ids = np.random.default_rng(12).integers(0, 49152, 2300).astype("<u4")          # five re-prime segments (the last one short)
    idf = tmp_path / "tok.ids"; ids.tofile(idf)
    inp = tmp_path / "in.txt"; inp.write_bytes(b"y" * 4000)
    one, two, back = tmp_path / "one.canz", tmp_path / "two.canz", tmp_path / "back.ids"
    base = ["-m", "candlezip_amd", "--synthetic", "1", "--token-ids", str(idf)]
    subprocess.run([sys.executable, *base, "compress", str(inp), str(one)], cwd=ROOT, check=True, capture_output=True)
    _ranks(2, [*base, "compress", str(inp), str(two)], 29111 + os.getpid() % 500)
    assert one.read_bytes() == two.read_bytes()
    _ranks(2, [*base, "decompress", str(two), str(back)], 28611 + os.getpid() % 500)   # rank 0 decodes the single stream
    assert np.array_equal(np.fromfile(back, "<u4"), ids)


def test_bench_two_ranks_rehearsal():
    """bench.py's N > 1 path (work list -> LPT share per rank, barrier, max-over-ranks timing, one JSON line from rank 0) with two gloo ranks
    on GPU 0 and the tiny model; the driver runs the RCCL / multi-GPU case"""
    argv = ["bench.py", "--gpus", "2", "--tiny", "--streams", "6", "--lanes", "2", "--lane-streams", "3", "--sample-tokens", "1536", "--window", "512",
            "--steps", "3", "--warmup", "1", "--profile-tokens", "1024", "--cpu-sample", "0"]
    procs = []
    port = 27611 + os.getpid() % 500
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CZ_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, *argv], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        outs.append(out)
    line = json.loads(outs[0].strip().splitlines()[-1])
    assert not any(l.startswith("{") for l in outs[1].splitlines())           # only rank 0 prints the line (gloo itself may chat on stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["generations_verified"] >= 1                       # 3 windows of 512 = one whole 1536-token sample: round trip checked
    assert "roofline" in line and line["config"]["streams_per_gpu"] == 6 and line["ranks_seen"] == 2 and line["windows_verified"] >= 3


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (no WORLD_SIZE in the environment) must not time one GPU and call it two: it starts
    two fresh child ranks itself before touching the GPU; rank 0's line says how many ranks the process group really had"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(CZ_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    argv = ["bench.py", "--gpus", "2", "--tiny", "--streams", "4", "--lanes", "2", "--lane-streams", "2", "--sample-tokens", "1024", "--window", "512",
            "--steps", "2", "--warmup", "0", "--profile-tokens", "0", "--cpu-sample", "0"]
    r = subprocess.run([sys.executable, *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dist_backend"] == "gloo" and line["value"] > 0
    assert line["windows_verified"] >= 2

"""bench.py --gpus N without a launcher (torchrun) around it starts its own N ranks: checked here on CPU with a stub rank script (the GPU
rehearsal is tests/test_gpu_multirank.py::test_bench_gpus_flag_spawns_its_own_ranks)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, n=3):
    stub = tmp_path / "rank.py"
    stub.write_text(textwrap.dedent(body))
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.exit(bench.spawn_ranks({n}, argv=['x'], script={str(stub)!r}))"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_spawn_ranks_env_and_single_line(tmp_path):
    r = _run(tmp_path, """
        import json, os
        print(json.dumps({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}))
    """)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                           # only rank 0's stdout is passed through
    env = json.loads(lines[0])
    assert env["RANK"] == "0" and env["WORLD_SIZE"] == "3" and env["MASTER_ADDR"] == "127.0.0.1" and int(env["MASTER_PORT"]) > 0
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_spawn_ranks_a_failing_rank_ends_the_job(tmp_path):
    r = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                                               # the others would wait at a barrier: they are terminated
    """)
    assert r.returncode == 7

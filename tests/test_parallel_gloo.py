"""§8e on CPU: the N>1 work distribution (no data-path collective) with world_size-2 gloo processes."""
import os
import sys
import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_assignment_is_balanced_and_complete():
    sys.path.insert(0, ROOT)
    from candlezip_amd.parallel import lpt_assign
    sizes = [34803, 39915, 4324, 41933, 17000, 17400, 100, 1]
    a = lpt_assign(sizes, 3)
    assert sorted(i for r in a for i in r) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(sizes)
    assert lpt_assign(sizes, 3) == a and lpt_assign([], 2) == [[], []]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import candlezip_amd as cz
    from candlezip_amd.parallel import run_sharded
    rng = np.random.default_rng(0)                                   # identical work list on every rank
    T = 1 << 30
    items = []
    for n in (500, 30, 1200, 7, 800, 64):
        cuts = np.sort(rng.integers(0, T, (n, 2)), axis=1).astype(np.uint32)
        cuts[:, 1] = np.maximum(cuts[:, 1], cuts[:, 0] + 1)
        items.append((cuts[:, 0].copy(), cuts[:, 1].copy()))

    def coder(batch):                                                # the per-stream serial coder (host side of the hot path)
        return [cz.ac_encode_pairs(lo, hi)[0] for lo, hi in batch]

    out = run_sharded(items, [len(i[0]) for i in items], coder, dist)
    ref = coder(items)
    ok = out == ref
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_sharded_streams_gather_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]

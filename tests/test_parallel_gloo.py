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


def test_segment_ranges_cover_whole_segments():
    sys.path.insert(0, ROOT)
    from candlezip_amd.parallel import segment_ranges
    for n, R, w in ((41933, 512, 8), (100, 512, 3), (0, 512, 2), (1024, 512, 2), (17408, 512, 5), (5000, 1024, 4)):
        rg = segment_ranges(n, R, w)
        assert len(rg) == w and rg[0][0] == 0 and rg[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rg, rg[1:]))                      # contiguous, in order
        assert all(b % R == 0 for b, _ in rg if b < n)                             # cut at re-prime boundaries only
        sizes = [(e - b + R - 1) // R for b, e in rg]
        assert max(sizes) - min(sizes) <= 1


def _shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import candlezip_amd as cz
    from candlezip_amd.parallel import encode_sharded
    ok = True
    for n, R in ((5000, 512), (300, 512), (1536, 512), (0, 512)):
        # a stand-in for the GPU evaluation of a token range: intervals that depend on the token index only (as the real ones depend on ids only)
        def enc(b, e):
            i = np.arange(b, e, dtype=np.uint64)
            lo = ((i * 2654435761) % (1 << 29)).astype(np.uint32)
            return lo, (lo + 1 + (i % 977).astype(np.uint32)).astype(np.uint32)
        lo, hi = encode_sharded(enc, n, R, dist)
        if rank == 0:
            rlo, rhi = enc(0, n)
            ok = ok and np.array_equal(lo, rlo) and np.array_equal(hi, rhi)
            ok = ok and cz.ac_encode_pairs(lo, hi)[0] == cz.ac_encode_pairs(rlo, rhi)[0]      # ONE payload, identical to the unsharded one
        else:
            ok = ok and lo is None and hi is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_single_file_segment_sharding_gathers_one_stream():
    """SURVEY §8e, single file: ranks evaluate blocks of whole re-prime segments, rank 0 gathers the pairs and runs the serial coder"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True), (2, True)]

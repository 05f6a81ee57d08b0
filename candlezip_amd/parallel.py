"""Work distribution across the GPUs of a node: files/chunks are independent .canz streams, so ranks never
exchange activations.  One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests) is used only to gather per-stream results and for barriers — there is no data-path collective."""
from typing import Callable, List, Sequence, Tuple

import numpy as np


def lpt_assign(sizes: Sequence[int], world: int) -> List[List[int]]:
    """longest-processing-time-first assignment of items (token counts) to ranks; deterministic, returns item indices per rank"""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += sizes[i]
    for r in range(world):
        out[r].sort()
    return out


def run_sharded(items: Sequence, sizes: Sequence[int], fn: Callable[[List], List], dist=None):
    """every rank calls this with the same `items`; rank r runs fn on its LPT share; all ranks get the full result list
    (results are small: payload bytes / ids per stream)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(fn(list(items)))
    world, rank = dist.get_world_size(), dist.get_rank()
    assign = lpt_assign(sizes, world)
    mine = fn([items[i] for i in assign[rank]])
    gathered = [None] * world
    dist.all_gather_object(gathered, list(zip(assign[rank], mine)))
    out = [None] * len(items)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out


# ---- ONE file over several GPUs (encode only: SURVEY §8e).  The reference's re-prime predicate (src/main.rs:2280) cuts the token loop into
# segments that depend on token ids alone, so each rank evaluates a block of whole segments (cz_encode_tokens_range_batch) and the
# (c_lo, c_hi) pairs — 8 bytes per token — are gathered on rank 0, which runs the serial arithmetic coder and writes the single .canz
# (src/main.rs:2358-2376).  Decode of one stream is sequential (symbol t+1 needs symbol t): it does not shard.
def segment_ranges(n_coded: int, reprime_interval: int, world: int) -> List[Tuple[int, int]]:
    """coded-token ranges [b, e) per rank: contiguous blocks of whole re-prime segments [k*R, (k+1)*R), as even as possible
    (blocks differ by at most one segment); ranks beyond the segment count get empty ranges"""
    R = max(int(reprime_interval), 1)
    nseg = (n_coded + R - 1) // R
    out = []
    for r in range(world):
        s0 = nseg * r // world; s1 = nseg * (r + 1) // world
        out.append((min(s0 * R, n_coded), min(s1 * R, n_coded)))
    return out


def encode_sharded(encode_range: Callable[[int, int], Tuple[np.ndarray, np.ndarray]], n_coded: int, reprime_interval: int, dist=None):
    """every rank calls this; rank r evaluates its block with encode_range(b, e) -> (c_lo, c_hi) of tokens [b, e); returns the full
    (c_lo, c_hi) on rank 0 and (None, None) elsewhere.  One gather of 8 B per token, outside the token loop."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return encode_range(0, n_coded)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = segment_ranges(n_coded, reprime_interval, world)
    b, e = ranges[rank]
    lo, hi = encode_range(b, e) if e > b else (np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    width = max(max(e1 - b1 for b1, e1 in ranges), 1)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(2, width, dtype=torch.int32)
    mine[0, : e - b] = torch.from_numpy(np.ascontiguousarray(lo, np.uint32).view(np.int32))
    mine[1, : e - b] = torch.from_numpy(np.ascontiguousarray(hi, np.uint32).view(np.int32))
    mine = mine.to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank != 0:
        return None, None
    full_lo = np.empty(n_coded, np.uint32); full_hi = np.empty(n_coded, np.uint32)
    for (b1, e1), t in zip(ranges, parts):
        a = t.cpu().numpy().view(np.uint32)
        full_lo[b1:e1] = a[0, : e1 - b1]; full_hi[b1:e1] = a[1, : e1 - b1]
    return full_lo, full_hi

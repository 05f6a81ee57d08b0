"""Work distribution across the GPUs of a node: files/chunks are independent .canz streams, so ranks never
exchange activations.  One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests) is used only to gather per-stream results and for barriers — there is no data-path collective."""
from typing import Callable, List, Sequence


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

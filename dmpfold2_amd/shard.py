"""Host-side sharding of independent alignments over the GPUs of one node.

The path shards embarrassingly: a target never needs more than one GPU and no state is shared
between targets, so each rank (one process per GPU) takes a subset of the
targets and predicts them; there is NO data-path collective.  `torch.distributed` is used only
for the result count / timing reductions of a job (backend "nccl" = RCCL on GPUs, "gloo" in the
CPU tests).
"""
from __future__ import annotations


def estimate_cost(L: int, N: int, iterations: int = 10) -> float:
    """Rough FLOP count of one prediction: pair trunk passes + vertical GRU
    + dense inverse.  Only used to balance shards."""
    n = min(int(N), 3000)
    passes = max(int(iterations), 0) + 1
    return passes * L * L * 5.3e7 + n * L * 4.8e6 + 2.0 * (21.0 * L) ** 3


def partition_targets(costs, world: int):
    """Longest-processing-time-first assignment of targets to `world` ranks.
    Returns a list (per rank) of target indices; deterministic, every index exactly once."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += float(costs[i])
    return shards


def job_summary(n_done: int, elapsed: float, group=None):
    """(total targets, max elapsed) over all ranks - the only communication of a sharded job."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return n_done, elapsed
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    cnt = torch.tensor([float(n_done)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    return int(round(cnt.item())), float(tmax.item())


def pin_rank_to_cores(local_rank: int, local_world: int) -> list:
    """One process per GPU on a node: every rank runs a polling scheduler thread (plus a helper).  When the
    process may run on every core of the machine - nobody has assigned cores yet - give rank r the r-th of
    `local_world` equal contiguous slices of the core list, so that eight schedulers do not migrate over each other.
    OPT-IN (DMP_PIN_CORES=1): which slice is close to which GPU depends on the node's topology, and no multi-GPU node
    was available to this build to measure it; the default leaves placement to the operating system.
    Returns the cores now allowed (unchanged if not opted in, the affinity was already restricted, the platform has no
    affinity call, or one rank only)."""
    import os
    if not hasattr(os, "sched_getaffinity"):
        return []
    cores = sorted(os.sched_getaffinity(0))
    if (local_world <= 1 or os.environ.get("DMP_PIN_CORES") != "1" or len(cores) != (os.cpu_count() or 0)
            or len(cores) < 2 * local_world):
        return cores
    per = len(cores) // local_world
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine

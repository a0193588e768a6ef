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


def job_summary(n_done: int, elapsed: float, group=None, failures=None):
    """(total targets, max elapsed) over all ranks - the only communication of a sharded job.  With `failures` = (failed
    targets of this rank, 1 if this rank itself broke down else 0) the answer also carries their sums over the ranks:
    (total, max elapsed, failed targets, broken ranks) - so that rank 0's summary shows that ANOTHER rank lost targets."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return (n_done, elapsed) if failures is None else (n_done, elapsed, int(failures[0]), int(failures[1]))
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    f = failures if failures is not None else (0, 0)
    cnt = torch.tensor([float(n_done), float(f[0]), float(f[1])], dtype=torch.float64, device=dev)
    tmax = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    if failures is None:
        return int(round(cnt[0].item())), float(tmax.item())
    return int(round(cnt[0].item())), float(tmax.item()), int(round(cnt[1].item())), int(round(cnt[2].item()))


def job_store(rank: int, world: int, timeout_s: float = 1800.0):
    """The key-value store of a multi-rank job, made with torch.distributed's PUBLIC constructor (the address of the
    launcher's environment: MASTER_ADDR / MASTER_PORT) so that it can be handed to init_process_group(store=...) AND
    to run_batch's shared work queue - no reach into the process group's private default store.  Under torchrun the
    launcher's agent already serves the store (TORCHELASTIC_USE_AGENT_STORE): every rank is a client then; otherwise
    rank 0 serves it."""
    import os
    from datetime import timedelta
    import torch.distributed as dist
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ["MASTER_PORT"])
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    return dist.TCPStore(host, port, world, is_master=(rank == 0 and not agent),
                         timeout=timedelta(seconds=timeout_s), multi_tenant=True)


def parse_cpulist(text: str) -> list:
    """'0-3,8,10-11' (the kernel's cpulist format) -> [0, 1, 2, 3, 8, 10, 11]."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cores(bdf: str, sysfs_root: str = "/sys") -> list:
    """Cores on the NUMA node the PCI device `bdf` ('0000:c1:00.0') hangs off: <sysfs>/bus/pci/devices/<bdf>/numa_node
    names the node (-1 = the platform does not say) and node<k>/cpulist its cores.  [] when it cannot be read."""
    import os
    try:
        node = int(open(os.path.join(sysfs_root, "bus/pci/devices", bdf, "numa_node")).read())
        if node < 0:
            return []
        return parse_cpulist(open(os.path.join(sysfs_root, "devices/system/node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return []


def plan_core_slices(local_cores, allowed) -> list:
    """Core slice per local rank.  `local_cores[r]` = the cores near rank r's GPU ([] = unknown), `allowed` = the cores
    this process may run on.  Ranks whose GPUs share a NUMA node split that node's (allowed) cores into equal
    contiguous parts in rank order; a rank whose node is unknown, or whose part would be empty, gets None (leave its
    placement to the operating system)."""
    allowed = set(allowed)
    by_node = {}
    for r, cores in enumerate(local_cores):
        key = tuple(c for c in cores if c in allowed)
        if key:
            by_node.setdefault(key, []).append(r)
    out = [None] * len(local_cores)
    for cores, ranks in by_node.items():
        per = len(cores) // len(ranks)
        if per < 2:                                    # a scheduler thread and its helper want a core each
            continue
        for k, r in enumerate(ranks):
            out[r] = list(cores[k * per:(k + 1) * per])
    return out


def device_bdf(index: int) -> str:
    """PCI address of HIP device `index` ('' if PyTorch does not expose it)."""
    import torch
    try:
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return ""


def pin_rank_to_cores(local_rank: int, local_world: int, bdfs=None, sysfs_root: str = "/sys") -> list:
    """One process per GPU on a node: every rank runs a scheduler thread (plus a helper).  When the process may run on
    every core of the machine - nobody has assigned cores yet - rank r is restricted to a slice of the cores of the
    NUMA node its GPU hangs off (`plan_core_slices` over the PCI addresses of the node's GPUs, read from sysfs): the
    DEFAULT whenever the topology can be read.  Where it cannot, nothing is changed unless DMP_PIN_CORES=1 asks for
    the r-th of `local_world` equal contiguous slices of the core list; DMP_PIN_CORES=0 switches pinning off.  No
    multi-GPU node was available to this build: the effect is unmeasured.
    Fewer than four cores per rank: nothing is changed (the HIP runtime's helper threads need room beside the scheduler).
    Returns the cores now allowed (unchanged if the affinity was already restricted, the platform has no affinity
    call, or one rank only)."""
    import os
    if not hasattr(os, "sched_getaffinity"):
        return []
    cores = sorted(os.sched_getaffinity(0))
    mode = os.environ.get("DMP_PIN_CORES", "")
    if local_world <= 1 or mode == "0" or len(cores) != (os.cpu_count() or 0) or len(cores) < 4 * local_world:
        return cores
    if bdfs is None:
        bdfs = [device_bdf(i) for i in range(local_world)]
    plan = plan_core_slices([gpu_local_cores(b, sysfs_root) if b else [] for b in bdfs], cores)
    mine = plan[local_rank] if local_rank < len(plan) else None
    if mine is None:
        if mode != "1":
            return cores
        per = len(cores) // local_world
        mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine

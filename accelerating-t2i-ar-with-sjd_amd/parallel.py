"""Prompt-parallel sharding + the single collective of the inference path.

The reference shards prompts over GPUs with independent processes and no communication
(dataset_tools/multi_gpu_dataframe_split.py:31-63, multi_gpu_infer_with_prompt.py:146-172).  Here: one process per GPU
under torch.distributed (backend "nccl" = RCCL over xGMI), prompt i -> rank by the same contiguous split, nothing
collective inside the decode loop, and ONE all_gather of [n_tokens, n_steps, seconds] (24 B per rank) at the end.
"""
import torch
import torch.distributed as dist


def contiguous_split(n_items: int, world: int, rank: int, scheme: str = "balanced"):
    """[lo, hi) of rank's contiguous shard of n_items prompts.
    "reference": the arithmetic of the reference's per-GPU dataframe split (dataset_tools/multi_gpu_dataframe_split.py:55-61): floor-sized
                 chunks, the LAST rank takes the remainder (15 prompts on 8 GPUs: 1,1,1,1,1,1,1,8).
    "balanced":  same contiguous order, shard sizes differ by at most one (15 on 8: 2,2,2,2,2,2,2,1) -- what the launcher uses: the job
                 ends with its slowest rank."""
    if scheme == "reference":
        per = n_items // world
        lo = rank * per
        return lo, (lo + per if rank < world - 1 else n_items)
    if scheme != "balanced":
        raise ValueError(f"split scheme {scheme!r}")
    per, extra = divmod(n_items, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def gpu_numa_cpus(pci_bus_id: str, sysfs: str = "/sys"):
    """CPUs of the NUMA node a GPU hangs off ("0000:c1:00.0" -> sorted cpu ids), or None when sysfs does not say (no NUMA, container)."""
    import os
    try:
        with open(os.path.join(sysfs, "bus/pci/devices", pci_bus_id.lower(), "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            spec = f.read().strip()
    except (OSError, ValueError):
        return None
    cpus = []
    for part in filter(None, spec.split(",")):
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return sorted(cpus) or None


def pin_to_gpu_numa_node(device_index: int, n_local_ranks: int = 1, local_rank: int = 0, sysfs: str = "/sys"):
    """One host loop per GPU, one sync every ~3 ms: keep each rank's loop (and its OpenMP / torch intra-op threads) on the cores next to
    ITS GPU so that eight of them do not collide on one socket (the reference leaves this to CUDA_VISIBLE_DEVICES + the OS,
    multi_gpu_infer_with_prompt.py:146-172).  Ranks that share a node split its cores evenly.  Best effort: returns the cpu list it pinned
    to, or None when the topology is not exposed (then nothing is changed)."""
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        bus = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None
    cpus = gpu_numa_cpus(bus, sysfs)
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    # ranks whose GPUs sit on the same node take disjoint slices of it
    same = [r for r in range(n_local_ranks) if _same_node(r, bus, sysfs)] if n_local_ranks > 1 else [local_rank]
    if local_rank in same and len(same) > 1 and len(allowed) >= len(same):
        k = len(allowed) // len(same)
        i = same.index(local_rank)
        allowed = allowed[i * k:(i + 1) * k]
    os.sched_setaffinity(0, allowed)
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(len(allowed), 16)))
    torch.set_num_threads(max(1, min(len(allowed), 16)))
    return allowed


def _same_node(other_index, bus, sysfs):
    try:
        p = torch.cuda.get_device_properties(other_index)
        ob = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return False
    return gpu_numa_cpus(ob, sysfs) == gpu_numa_cpus(bus, sysfs)


def run_prompt_queue(n_prompts: int, decode_one, sync=None, device=None, scheme: str = "balanced"):
    """The reference's fan-out (M prompts over N single-GPU processes, dataset_tools/multi_gpu_infer_with_prompt.py:146-172) on one
    process group: this rank decodes prompts contiguous_split(M, world, rank) one after the other -- decode_one(i) -> (tokens, steps);
    engine, graphs and cache rows are the caller's and are reused from prompt to prompt -- between a barrier and ONE all_gather of
    [tokens, steps, seconds].  No collective inside.  Returns dict(per_rank=[(tokens, steps, seconds)], shard=(lo, hi), **aggregate)."""
    import time
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    lo, hi = contiguous_split(n_prompts, world, rank, scheme)
    if dist.is_available() and dist.is_initialized():          # (also at world size 1: the same path the 8-GPU launch takes)
        dist.barrier()
    if sync is not None:
        sync()
    t0 = time.perf_counter()
    tokens = steps = 0
    for i in range(lo, hi):
        t, s = decode_one(i)
        tokens, steps = tokens + int(t), steps + int(s)
    if sync is not None:
        sync()
    seconds = time.perf_counter() - t0
    rep = gather_report(tokens, steps, seconds, device)
    out = aggregate(rep)
    out.update(per_rank=rep, shard=(lo, hi), world=world, prompts=n_prompts)
    return out


def gather_report(n_tokens, n_steps, seconds, device=None, extra=()):
    """-> list over ranks of (n_tokens, n_steps, seconds, *extra).  One all_gather; no-op without a process group."""
    vals = [float(n_tokens), float(n_steps), float(seconds)] + [float(x) for x in extra]
    if not (dist.is_available() and dist.is_initialized()):
        return [tuple(vals)]
    backend = dist.get_backend()
    dev = device if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor(vals, dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [tuple(t.tolist()) for t in out]


def aggregate(report):
    """whole-job throughput: all ranks' tokens over the slowest rank's time."""
    tokens = sum(r[0] for r in report)
    steps = sum(r[1] for r in report)
    t = max(r[2] for r in report)
    return dict(tokens=tokens, steps=steps, seconds=t, tokens_per_s=tokens / t if t > 0 else 0.0,
                tokens_per_step=tokens / steps if steps else 0.0)

"""Prompt-parallel sharding + the single collective of the inference path.

The reference shards prompts over GPUs with independent processes and no communication
(dataset_tools/multi_gpu_dataframe_split.py:31-63, multi_gpu_infer_with_prompt.py:146-172).  Here: one process per GPU
under torch.distributed (backend "nccl" = RCCL over xGMI), prompt i -> rank by the same contiguous split, nothing
collective inside the decode loop, and ONE all_gather of [n_tokens, n_steps, seconds] (24 B per rank) at the end.
"""
import torch
import torch.distributed as dist


def contiguous_split(n_items: int, world: int, rank: int):
    """[lo, hi) of rank's shard; same arithmetic as the reference's per-GPU dataframe split: ceil-sized chunks,
    the last ranks may get fewer / no items."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def gather_report(n_tokens, n_steps, seconds, device=None):
    """-> list over ranks of (n_tokens, n_steps, seconds).  One all_gather; no-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [(float(n_tokens), float(n_steps), float(seconds))]
    backend = dist.get_backend()
    dev = device if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(n_tokens), float(n_steps), float(seconds)], dtype=torch.float64, device=dev)
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

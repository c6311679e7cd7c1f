"""CPU, world_size 2 over gloo: the prompt-parallel path (contiguous prompt split + the single all_gather report).
On the GPU node the same code runs with backend "nccl" (= RCCL over xGMI); nothing else in the decode loop is collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sjd_amd.parallel import aggregate, contiguous_split, gather_report


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_prompts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = contiguous_split(n_prompts, world, rank)
    # stand-in decode: prompt i "emits" 100+i tokens in 10+i steps and takes (i+1) seconds
    tokens = sum(100 + i for i in range(lo, hi))
    steps = sum(10 + i for i in range(lo, hi))
    seconds = float(sum(i + 1 for i in range(lo, hi)))
    dist.barrier()
    rep = gather_report(tokens, steps, seconds)
    q.put((rank, (lo, hi), rep))
    dist.barrier()
    dist.destroy_process_group()


def test_contiguous_split_covers_all_prompts():
    for n in (0, 1, 7, 8, 9, 30):
        for world in (1, 2, 4, 8):
            parts = [contiguous_split(n, world, r) for r in range(world)]
            flat = [i for lo, hi in parts for i in range(lo, hi)]
            assert flat == list(range(n))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= (n + world - 1) // world


def test_gather_report_world2_gloo():
    world, n_prompts, port = 2, 5, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_prompts, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, rep0), (r1, s1, rep1) = out
    assert (s0, s1) == ((0, 3), (3, 5))
    assert rep0 == rep1                                   # every rank sees the same gathered report
    assert rep0 == [(303.0, 33.0, 6.0), (207.0, 27.0, 9.0)]
    agg = aggregate(rep0)
    assert agg["tokens"] == 510 and agg["seconds"] == 9.0 and abs(agg["tokens_per_s"] - 510 / 9.0) < 1e-9


def test_gather_report_without_process_group():
    assert gather_report(5, 2, 1.5) == [(5.0, 2.0, 1.5)]

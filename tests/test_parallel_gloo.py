"""CPU, world_size 2 over gloo: the prompt-parallel path (contiguous prompt split + the single all_gather report).
On the GPU node the same code runs with backend "nccl" (= RCCL over xGMI); nothing else in the decode loop is collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sjd_amd.parallel import aggregate, contiguous_split, gather_report, gpu_numa_cpus, run_prompt_queue


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
    for scheme in ("balanced", "reference"):
        for n in (0, 1, 7, 8, 9, 15, 30):
            for world in (1, 2, 4, 8):
                parts = [contiguous_split(n, world, r, scheme) for r in range(world)]
                flat = [i for lo, hi in parts for i in range(lo, hi)]
                assert flat == list(range(n))
                if scheme == "balanced":
                    assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1
    # the reference's arithmetic (dataset_tools/multi_gpu_dataframe_split.py:55-61): floor-sized chunks, the last GPU takes the remainder
    assert [contiguous_split(15, 8, r, "reference") for r in range(8)] == [(i, i + 1) for i in range(7)] + [(7, 15)]
    assert [contiguous_split(15, 8, r) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12), (12, 14), (14, 15)]


def test_gpu_numa_cpus_reads_sysfs(tmp_path):
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("3\n")
    node = tmp_path / "devices/system/node/node3"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("48-51,176-177\n")
    assert gpu_numa_cpus("0000:C1:00.0", str(tmp_path)) == [48, 49, 50, 51, 176, 177]
    (dev / "numa_node").write_text("-1\n")
    assert gpu_numa_cpus("0000:c1:00.0", str(tmp_path)) is None
    assert gpu_numa_cpus("0000:99:00.0", str(tmp_path)) is None


def _queue_worker(rank, world, port, n_prompts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def decode_one(i):             # stub engine: prompt i emits 100 + i tokens in 10 + i steps
        seen.append(i)
        return 100 + i, 10 + i

    out = run_prompt_queue(n_prompts, decode_one)
    q.put((rank, seen, out["shard"], out["tokens"], out["steps"], [r[:2] for r in out["per_rank"]], out["seconds"] >= max(r[2] for r in out["per_rank"]) - 1e-12))
    dist.barrier()
    dist.destroy_process_group()


def test_prompt_queue_world2_gloo():
    """M = 5 prompts over 2 ranks: contiguous shards, every prompt decoded exactly once, ONE gathered report that both ranks agree on"""
    world, n_prompts, port = 2, 5, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue_worker, args=(r, world, port, n_prompts, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seen0, shard0, tok0, st0, rep0, ok0), (r1, seen1, shard1, tok1, st1, rep1, ok1) = out
    assert (seen0, seen1) == ([0, 1, 2], [3, 4]) and (shard0, shard1) == ((0, 3), (3, 5))
    assert tok0 == tok1 == sum(100 + i for i in range(5)) and st0 == st1 == sum(10 + i for i in range(5))
    assert rep0 == rep1 == [(303.0, 33.0), (207.0, 27.0)] and ok0 and ok1


def _queue8_worker(rank, world, port, n_prompts, scheme, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def decode_one(i):             # stub engine: prompt i emits 100 + i tokens in 10 + i steps; odd ranks are slower
        seen.append(i)
        if rank % 2:
            import time
            time.sleep(0.02)
        return 100 + i, 10 + i

    out = run_prompt_queue(n_prompts, decode_one, scheme=scheme)
    q.put((rank, seen, out["shard"], out["tokens"], out["steps"], [r[:2] for r in out["per_rank"]], out["world"],
           out["seconds"] >= max(r[2] for r in out["per_rank"]) - 1e-12))
    dist.barrier()
    dist.destroy_process_group()


def _run_world8(n_prompts, scheme):
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue8_worker, args=(r, world, port, n_prompts, scheme, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_prompt_queue_world8_gloo_both_split_schemes():
    """config 4's shape without the hardware: 8 ranks, 15 prompts (dataset_tools/multi_gpu_infer_with_prompt.py:146-172,
    multi_gpu_dataframe_split.py:47-63) -- contiguous shards under both split schemes, every prompt decoded exactly once, ONE gathered report
    all eight ranks agree on, the job's time = its slowest rank's."""
    for scheme, sizes in (("balanced", [2, 2, 2, 2, 2, 2, 2, 1]), ("reference", [1, 1, 1, 1, 1, 1, 1, 8])):
        out = _run_world8(15, scheme)
        assert [len(o[1]) for o in out] == sizes
        assert [i for o in out for i in o[1]] == list(range(15))                 # contiguous, in rank order, nothing twice
        assert all(o[2] == contiguous_split(15, 8, o[0], scheme) for o in out)
        assert all(o[3] == sum(100 + i for i in range(15)) and o[4] == sum(10 + i for i in range(15)) and o[6] == 8 and o[7] for o in out)
        assert all(o[5] == out[0][5] for o in out)                               # the same per-rank report on every rank
        assert out[0][5] == [(float(sum(100 + i for i in range(lo, hi))), float(sum(10 + i for i in range(lo, hi))))
                             for lo, hi in (contiguous_split(15, 8, r, scheme) for r in range(8))]


def test_prompt_queue_world8_gloo_ranks_without_prompts():
    """fewer prompts than ranks: five ranks own an empty shard, finish at once and must still meet the others at the one all_gather"""
    out = _run_world8(3, "balanced")
    assert [o[1] for o in out] == [[0], [1], [2], [], [], [], [], []]
    assert all(o[3] == 303 and o[4] == 33 for o in out) and all(o[5] == out[0][5] for o in out)
    assert out[0][5][3:] == [(0.0, 0.0)] * 5


def test_prompt_queue_without_process_group():
    out = run_prompt_queue(3, lambda i: (10, 2))
    assert out["tokens"] == 30 and out["steps"] == 6 and out["shard"] == (0, 3) and out["world"] == 1


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

"""GPU, 2+ devices: BASELINE.json config 4's only collective -- the RCCL all_gather of the per-rank [tokens, steps, seconds] report --
run for real over backend "nccl" (RCCL/xGMI) with two ranks, and `bench.py --gpus 2` end to end on a small model (skipped on a
1-GPU box; the driver's 8-GPU scaling run exercises the same path) -- and, on ANY box, the same init / barrier / all_gather path at world
size 1 on the real RCCL library: `gather_report` on backend "nccl" and `SJD_FORCE_DIST=1 bench.py --total-prompts 3`
(reference fan-out: dataset_tools/multi_gpu_infer_with_prompt.py:146-172)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from sjd_amd.parallel import contiguous_split, gather_report
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lo, hi = contiguous_split(5, world, rank)
    rep = gather_report(sum(100 + i for i in range(lo, hi)), sum(10 + i for i in range(lo, hi)), float(hi - lo), dev)
    q.put((rank, rep))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")


def test_gather_report_world2_rccl(two_gpus):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1] == [(303.0, 33.0, 3.0), (207.0, 27.0, 2.0)]


def test_bench_gpus_2_prints_one_line_with_n_gpus_2(two_gpus):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "lumina_tiny", "--steps", "12", "--warmup", "3",
           "--kv-center", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["prompts"] == 2 and line["steps"] == 12 and line["scaling"] == "weak"


def _worker_world1(port, q):
    import torch.distributed as dist
    from sjd_amd.parallel import gather_report
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    dist.barrier()
    rep = gather_report(321, 45, 6.5, dev, extra=(7.0, 8.0, 9.0))
    t = torch.arange(6, dtype=torch.float64, device=dev)
    outs = [torch.empty_like(t)]
    dist.all_gather(outs, t)                      # the collective itself, on the RCCL communicator
    q.put((rep, outs[0].cpu().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_report_world1_on_rccl():
    """ONE RCCL call on the box the driver has: init_process_group("nccl") + barrier + the all_gather of the step report, world size 1."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    p = ctx.Process(target=_worker_world1, args=(port, q))
    p.start()
    rep, gathered = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert rep == [(321.0, 45.0, 6.5, 7.0, 8.0, 9.0)] and gathered == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]


def test_bench_prompt_queue_world1_over_rccl():
    """`SJD_FORCE_DIST=1 python bench.py --total-prompts 3` -- config 4's launcher at world size 1 with the process group on backend
    "nccl": init, the barrier brackets, the prompt queue and the ONE all_gather all run on the real RCCL library."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, SJD_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--model", "lumina_tiny", "--total-prompts", "3", "--steps", "8", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["prompts"] == 3 and len(line["per_rank"]) == 1
    assert line["per_rank"][0]["tokens"] > 0 and len(line["rank0_prompts"]) == 3

"""GPU, 2+ devices: BASELINE.json config 4's only collective -- the RCCL all_gather of the per-rank [tokens, steps, seconds] report --
run for real over backend "nccl" (RCCL/xGMI) with two ranks, and `bench.py --gpus 2` end to end on a small model.  Skipped on a
1-GPU box (the driver's 8-GPU scaling run exercises the same path)."""
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

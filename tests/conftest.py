import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def poison_device_memory(total_gib=24):
    """Fill the caching allocator's free pool with 0xFF bytes (NaN as fp32 / fp16 / bf16, -1 as an integer): a later torch.empty() hands out
    poisoned blocks, so a kernel that reads workspace nobody wrote fails in EVERY test order instead of only after an unlucky predecessor.
    A fresh process gets zero pages from the driver, which hides exactly that class of defect."""
    import torch
    if not torch.cuda.is_available():
        return
    torch.cuda.synchronize()
    # scaled to the device: never more than a quarter of what is free right now -- the extension's own hipMalloc'd buffers,
    # hipBLASLt workspaces and a neighbour on a shared GPU allocate OUTSIDE the caching allocator and must still find room
    free_b, _ = torch.cuda.mem_get_info()
    total_gib = max(1, min(int(total_gib), int(free_b // 4) >> 30))
    held = []
    try:
        for nbytes, count in ((1 << 30, max(0, total_gib - 8 if total_gib > 10 else total_gib - 2)), (64 << 20, 64 if total_gib >= 8 else 8),
                              (8 << 20, 256 if total_gib >= 8 else 64), (1 << 20, 1024), (64 << 10, 2048), (4 << 10, 2048)):
            for _ in range(count):
                held.append(torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda:0"))
    except torch.OutOfMemoryError:
        pass
    torch.cuda.synchronize()
    del held          # back to the allocator's pool, contents kept


# modules whose tests launch kernels that read workspace a previous launch must have written (K1 split partials, G1 split-K planes, head
# partials): poisoned BY DEFAULT, so that the driver's fresh-box run -- where the allocator hands out zero pages -- can see an unwritten
# partial too (GPUTEST_r03 was green on a K1 that merged stale workspace)
_POISON_BY_DEFAULT = ("test_gpu_kernels", "test_gpu_real_shape_forward", "test_gpu_glue", "test_gpu_fp8_model_bound")


_POISON_COUNT = {}


@pytest.fixture(autouse=True)
def _poison_before_gpu_tests(request):
    """Poison the allocator's free pool before a gpu test.  Default: the K1 / workspace modules above -- the FULL 8 GiB pool before a module's
    first test and before every 16th after it, a light refresh (the small size classes a test's torch.empty() calls are served from, ~0.3 GiB)
    before the others: the full fill in front of every test cost the suite minutes.  SJD_TEST_POISON=1: the full 24 GiB fill before
    every gpu test (the poisoned run of the round, profiles/r*_full_gpu_suite_poisoned.txt); SJD_TEST_POISON=0: never -- for small or shared GPUs."""
    mode = os.environ.get("SJD_TEST_POISON", "")
    if mode != "0" and request.node.get_closest_marker("gpu") is not None:
        mod = request.node.module.__name__.rsplit(".", 1)[-1]
        if mode == "1":
            poison_device_memory()
        elif mod in _POISON_BY_DEFAULT:
            n = _POISON_COUNT[mod] = _POISON_COUNT.get(mod, -1) + 1
            poison_device_memory(total_gib=8 if n % 16 == 0 else 1)
    yield

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
    held = []
    try:
        for nbytes, count in ((1 << 30, max(2, total_gib - 8)), (64 << 20, 64), (8 << 20, 256), (1 << 20, 1024), (64 << 10, 2048), (4 << 10, 2048)):
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


@pytest.fixture(autouse=True)
def _poison_before_gpu_tests(request):
    """Poison the allocator's free pool before a gpu test.  Default: the K1 / workspace modules above, with an 8 GiB pool (the free blocks a test's
    torch.empty() calls are served from).  SJD_TEST_POISON=1: every gpu test, 24 GiB (tools/profile_round.sh runs the suite once this way);
    SJD_TEST_POISON=0: never."""
    mode = os.environ.get("SJD_TEST_POISON", "")
    if mode != "0" and request.node.get_closest_marker("gpu") is not None:
        if mode == "1":
            poison_device_memory()
        elif request.node.module.__name__.rsplit(".", 1)[-1] in _POISON_BY_DEFAULT:
            poison_device_memory(total_gib=8)
    yield

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
        for nbytes, count in ((1 << 30, max(1, total_gib - 8)), (64 << 20, 64), (8 << 20, 256), (1 << 20, 1024), (64 << 10, 2048), (4 << 10, 2048)):
            for _ in range(count):
                held.append(torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda:0"))
    except torch.OutOfMemoryError:
        pass
    torch.cuda.synchronize()
    del held          # back to the allocator's pool, contents kept


@pytest.fixture(autouse=True)
def _poison_before_gpu_tests(request):
    """SJD_TEST_POISON=1: poison the allocator's free pool before every gpu test (tools/_r4_suite.sh runs the suite once this way)."""
    if os.environ.get("SJD_TEST_POISON") == "1" and request.node.get_closest_marker("gpu") is not None:
        poison_device_memory()
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def usable_cores():
    """Host threads this process may really use: affinity clipped by the cgroup CPU quota (the GPU box shows 256 cores
    behind a 16-CPU quota; torch's default of one thread per visible core makes the oracle crawl there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


@pytest.fixture(autouse=True, scope="session")
def _torch_threads():
    import torch
    torch.set_num_threads(usable_cores())
    yield

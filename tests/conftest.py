import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _gpu_tests_need_a_gpu(request):
    """`-m gpu` tests are the parity tests proper and run on a GPU box; a plain `pytest` on a CPU box skips them
    (the product path itself has no CPU fallback and raises)."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if not torch.cuda.is_available():
            pytest.skip("needs an MI355X")


def usable_cpus():
    """CPUs this process may really use: affinity capped by the cgroup quota (a GPU box shows 256 CPUs and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


@pytest.fixture(scope="session", autouse=True)
def _torch_threads_fit_the_cgroup():
    """The float64 torch-CPU checkers (oracle/torch_train.py, oracle/torch_port.py) run 10x slower when torch starts one thread
    per visible CPU on a box whose cgroup grants a sixteenth of them."""
    import torch
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus())))
    yield


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load

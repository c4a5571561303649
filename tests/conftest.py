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

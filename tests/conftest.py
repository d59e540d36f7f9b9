import os
import sys

import pytest

# the process-wide switches of the C ABI (ac_gemm_set_*, ac_set_persistent_kernels) are test hooks: inert unless the process asks
# for them before libacamd.so is loaded (include/acamd.h "TEST HOOKS")
os.environ.setdefault("AC_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "adaptive-classifier_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda_dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

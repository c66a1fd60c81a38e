"""pytest configuration: the `gpu` marker and shared paths.

`-m "not gpu"` runs on the CPU-only build container (oracle vs golden fixtures, host logic, ABI
surface, 2-process gloo data-parallel path); `-m gpu` runs on an MI355X and calls the HIP library
through its C ABI.  Nothing here reads /root/reference.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(GOLDEN / f"{name}.npz") as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


@pytest.fixture(scope="session")
def cuda_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")

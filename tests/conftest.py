import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity tests run with the built-in tile heuristics: the first-use autotuner may pin a different wgrad tile from run to
# run (a different split-K summation order), which the gradient yardsticks of the deep-network tests would see as noise.
# tests/test_gpu_kernels.py::test_tile_menu_* covers the pinned tiles explicitly.
os.environ.setdefault("SUHPE_AUTOTUNE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # The CPU oracle is the slow half of most GPU tests.  torch's default thread pool = every hardware thread of the host (256 on the
    # GPU box), where the oracle's small convolutions run 2-3x SLOWER than on 32 threads (measured for bench.py's cpu_baseline): cap it.
    import torch
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(ncores, 32)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

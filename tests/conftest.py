import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is most of the GPU suite's wall time: give torch as many threads as the container's CPU QUOTA holds
    # (16 on the GPU boxes), not as many as it sees (256 CPUs -> 128 threads -> 7.7 x slower; oracle.host_cpus)
    import torch
    from oracle import host_cpus
    torch.set_num_threads(host_cpus())
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cpus()))      # child processes of the multi-rank / kernel-variant tests


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch conv2d / autograd on the host) runs inside most GPU tests.  The GPU box grants a 16-core share of a
    # much larger host; torch's default of one thread per visible core oversubscribes that share (measured: 24 ms per tiny
    # conv2d, 30 s per lifting-gradient case, against 2.5 s with a sane thread count), so cap the intra-op threads.
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(8, n)))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")

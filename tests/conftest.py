import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The oracle runs on torch's OpenMP pool.  On a shared host a preempted worker leaves the others spinning at every barrier (seen
# here: the tiny-UNet loop tests 9 s -> 600 s while another tenant was busy); a short spin before sleeping (instead of libgomp's
# default 300 000 iterations) keeps the suite's time bounded at no cost on an idle host.  Has to
# be in the environment before the OpenMP runtime loads, i.e. before the first `import torch` of the session.
os.environ.setdefault("GOMP_SPINCOUNT", "30000")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_sessionstart(session):
    # the GPU box reports 256 CPUs but grants ~16: an oversubscribed torch CPU pool makes the oracle 10x slower
    import torch
    torch.set_num_threads(_usable_cpus())

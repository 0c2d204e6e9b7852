import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suites measure the library as a host gets it: a stray SRACK_* tuning variable in the environment (tools/ set them for experiments;
    the library reads them: other chunk lengths, generator variants ...) would test something else.  srack_render_info lists such variables
    ("knobs=[...]"); here the run refuses to start.  (Where the kernel cache lives and a test's own hooks are not tuning knobs.)"""
    stray = sorted(k for k in os.environ if k.startswith("SRACK_") and not k.startswith(("SRACK_KERNEL_CACHE_", "SRACK_BENCH_", "SRACK_TEST_")))
    if stray:
        raise pytest.UsageError("SRACK_* tuning variables are set: " + ", ".join(stray) + " — unset them for the test suites")


@pytest.fixture(scope="session")
def W():
    import srack_pkg
    return srack_pkg.load_workloads()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    import dentist_amd
    ctx = dentist_amd.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def cfg2_workload():
    """BASELINE configs[2] / [3]: 100 Mb assembly, 1 000 gaps, 1 M x 15 kb reads at 13 % (15.7 Gbp) -- built
    once per session (about a minute of host time), shared by the full-size tests."""
    from dentist_amd import sim
    return sim.Workload(100_000_000, 1000, 1_000_000, 15_000, seed=20260929)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle_c():
    """The C restatement (oracle/wgsl_oracle.c), built on demand.  Test infrastructure only."""
    from oracle import wgsl_oracle as wo
    return wo.CLib()


@pytest.fixture(scope="session")
def gpu():
    """GpuInstance::new().  GPU tests must run on the HIP kernels or fail: no skip, no fallback."""
    # created before any test imports torch: the library binds to the one HIP runtime loaded at that point
    import wgmath_amd as wg
    inst = wg.GpuInstance.new()
    yield inst
    inst.sync()

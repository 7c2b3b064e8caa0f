"""pytest configuration: registers the `gpu` marker and makes the repo root importable.

`-m "not gpu"` : oracle vs the reference's test identities / golden fixtures, host logic, C-ABI
                 symbol export -- runs on a CPU-only box in a few minutes.
`-m gpu`       : parity tests proper -- the HIP path through the C ABI vs the oracle, on an MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The HIP runtime aborts the process on a GPU fault or a hardware exception event -- also one raised by ANOTHER tenant of a shared host resetting
# the GPUs -- and says why only at log level 1 and above (errors); glibc's fatal messages go to the terminal unless told otherwise.  Make both
# end up in the test log (before the runtime is loaded).
os.environ.setdefault("AMD_LOG_LEVEL", "1")
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def Jets():
    """The product package, initialised on cuda:0.  GPU tests only."""
    import jets_jl_amd as J

    J.init(0)
    return J


@pytest.fixture(scope="session")
def oracle():
    from oracle import jets_oracle

    return jets_oracle

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure).  Built on demand with gcc/g++."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def native_lib():
    """The product library.  Built on demand (hipcc cross-compiles without a GPU)."""
    from meryl_amd import build, capi
    build.build()
    return capi.lib()


def golden_cases():
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "count_cases.json")
    with open(path) as f:
        return json.load(f)["cases"]

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libnthash_ref.so not built (reference tree absent)")
    return Reference()


@pytest.fixture(scope="session")
def built_lib():
    """Build (if needed) and load the C-ABI library: loud failure if it cannot be built."""
    from nthash_amd import build as nb
    nb.build()
    import nthash_amd
    return nthash_amd.load()


@pytest.fixture(scope="session")
def ctx(built_lib):
    import nthash_amd
    c = nthash_amd.Context(0)  # raises NtHipError(NODEVICE) without a GPU: gpu tests only
    yield c
    c.close()

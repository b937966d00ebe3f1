import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The C++ facade rolls sequences of up to 32768 bases on the host (a device round trip per object would cost more than
# the hashing); the GPU tests are about the device path, so they send every sequence there.  One test re-runs the
# facade suite in the default mode (tests/test_gpu_facade.py::test_facade_suite_again_*).
os.environ.setdefault("NTHASH_AMD_FORCE_DEVICE", "1")


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


@pytest.fixture(scope="session")
def facade(built_lib):
    """oracle/ref_shim.cpp (the driver that also runs the REAL reference when fixtures are
    generated) compiled against nthash_amd's own header + libnthash.so: drives our C++ facade."""
    import subprocess

    from oracle.pyoracle import Reference
    lib = os.path.join(ROOT, "nthash_amd", "lib")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfacade_shim.so")
    src = os.path.join(ROOT, "oracle", "ref_shim.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", f"-I{os.path.join(ROOT, 'include')}",
                           src, "-o", so, f"-L{lib}", "-lnthash", "-lnthash_hip", f"-Wl,-rpath,{lib}"])
    return Reference(so_path=so)

"""nthash_amd -- MI355X-native (gfx950) rolling k-mer hash engine, bit-exact with
bcgsc/ntHash 2.4.0 (ntHash_v2).

The product is the C-ABI shared library nthash_amd/lib/libnthash_hip.so
(include/nthash_hip.h) and the C++ host facade libnthash.so
(include/nthash/nthash.hpp).  This Python package is only the loader used by the
tests and bench.py; importing it never falls back to a CPU implementation.
"""
from . import capi  # noqa: F401
from .capi import Context, Multi, NtHipError, Seeds, device_count, load  # noqa: F401

__all__ = ["capi", "Context", "Multi", "NtHipError", "Seeds", "device_count", "load"]

"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/nthash_hip.h declares; without a GPU it fails
loudly instead of computing anything on the CPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "nthash_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nthip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for must in ("nthip_kmer_hash", "nthip_seed_hash", "nthip_seeds_create", "nthip_ctx_create"):
        assert must in syms


def test_library_exports_every_declared_symbol(built_lib):
    from nthash_amd import capi
    raw = ctypes.CDLL(capi.LIB_PATH)
    declared = header_symbols()
    assert declared, "no symbols parsed from the header"
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/nthash_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == declared, "nthash_amd/capi.py is out of sync with the header"


def test_version_and_error_strings(built_lib):
    assert b"gfx950" in built_lib.nthip_version()
    assert isinstance(built_lib.nthip_last_error(), bytes)


def test_no_cpu_fallback_without_device(built_lib):
    import nthash_amd
    if nthash_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nthash_amd.NtHipError) as ei:
        nthash_amd.Context(0)
    assert ei.value.code == nthash_amd.capi.NTHIP_ERR_NODEVICE


def test_product_does_not_reference_oracle():
    """nothing under nthash_amd/ or include/ may import, link or mention oracle/"""
    bad = []
    for top in ("nthash_amd", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".so", ".pyc", ".o")):
                    continue
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"oracle[/.]|nthash_oracle|liboracle|pyoracle|_ref/", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad

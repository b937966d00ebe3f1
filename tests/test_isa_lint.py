"""The ISA lint of the hidden-load construct (nthash_amd/isa_lint.py, run by nthash_amd/build.py on every unit).

Nine sites in five kernels issue the next tile's global loads in inline asm and consume them behind a counted
s_waitcnt; hipcc does not know the registers are not ready.  Round 2 guarded that with "no kernel spills"; a copy or a
use of such a register before the wait passes that check.  These tests pin (a) the lint's rules on hand-written
assembly, (b) that the build's reports cover every site of every instantiation and are clean, and (c) that a
deliberately broken build (-DNT_LINT_SELFTEST=1: a v_mov of a loaded register between the load and its wait) is refused.
"""
import glob
import json
import os
import subprocess

import pytest

from conftest import ROOT
from nthash_amd import isa_lint


def kernel(body):
    return "\t.text\nkern:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"


GOOD = kernel("""
.LBB0_1:
	;;#ASMSTART
	global_load_dword v9, v[4:5], off sc1
	global_load_dwordx4 v[0:3], v[6:7], off nt
	;;#ASMEND
	v_add_u32_e32 v10, v11, v12
	global_store_dwordx4 v[20:21], v[12:15], off
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	;;#ASMSTART
	; NTLINT_CONSUME v[0:3] v9
	;;#ASMEND
	v_mov_b32_e32 v30, v0
	s_cbranch_scc1 .LBB0_1
""")


def test_clean_site_passes():
    rep = isa_lint.lint_text(GOOD)
    assert rep["violations"] == []
    assert rep["hidden_loads"] == 2 and rep["consume_markers"] == 1
    assert rep["kernels"]["kern"]["counted_waits"] == [{"line": 12, "vmcnt": 1}]


def test_use_before_the_wait_is_refused():
    bad = GOOD.replace("v_add_u32_e32 v10, v11, v12", "v_add_u32_e32 v10, v2, v12")
    v = isa_lint.lint_text(bad)["violations"]
    assert len(v) == 1 and "touches v[2]" in v[0]


def test_overwrite_before_the_wait_is_refused():
    bad = GOOD.replace("v_add_u32_e32 v10, v11, v12", "v_mov_b32_e32 v9, 0")
    v = isa_lint.lint_text(bad)["violations"]
    assert len(v) == 1 and "touches v[9]" in v[0]


def test_copy_between_load_and_wait_is_refused():
    """what hipcc did to the vmcnt(0) branch of the headline kernel's wait until round 3: the loaded registers copied
    BEFORE the wait, the marker naming the copies"""
    bad = GOOD.replace("\tv_add_u32_e32 v10, v11, v12\n", "\tv_mov_b32_e32 v40, v9\n").replace(
        "NTLINT_CONSUME v[0:3] v9", "NTLINT_CONSUME v[0:3] v40")
    v = isa_lint.lint_text(bad)["violations"]
    assert any("copied between its load and the wait" in x for x in v)
    assert any("touches v[9]" in x for x in v)


def test_marker_without_a_wait_is_refused():
    bad = GOOD.replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n", "")
    v = isa_lint.lint_text(bad)["violations"]
    assert len(v) == 1 and "without an inline s_waitcnt" in v[0]


def test_compiler_made_counted_wait_does_not_count():
    bad = GOOD.replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n", "\ts_waitcnt vmcnt(1)\n")
    assert any("without an inline s_waitcnt" in x for x in isa_lint.lint_text(bad)["violations"])
    ok = GOOD.replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)\n\t;;#ASMEND\n", "\ts_waitcnt vmcnt(0)\n")
    assert isa_lint.lint_text(ok)["violations"] == []  # everything has landed, whoever asked for it


def test_path_that_skips_the_wait_is_refused():
    bad = GOOD.replace("\tglobal_store_dwordx4 v[20:21], v[12:15], off\n",
                       "\tglobal_store_dwordx4 v[20:21], v[12:15], off\n\ts_cbranch_vccz .LBB0_2\n").replace(
        "\t;;#ASMSTART\n\t; NTLINT_CONSUME", ".LBB0_2:\n\t;;#ASMSTART\n\t; NTLINT_CONSUME")
    v = isa_lint.lint_text(bad)["violations"]
    assert any("without an inline s_waitcnt" in x for x in v)


def test_touch_on_a_loop_exit_path_is_refused_unless_everything_was_waited_for():
    tail = "\tv_mov_b32_e32 v0, 0\n\tglobal_atomic_add_x2 v0, v[26:27], s[0:1]\n"
    bad = GOOD.replace("\tv_add_u32_e32 v10, v11, v12\n", "\ts_cbranch_vccz .LBB0_9\n\tv_add_u32_e32 v10, v11, v12\n").replace(
        "\ts_endpgm\n", ".LBB0_9:\n" + tail + "\ts_endpgm\n")
    assert any("touches v[0]" in x for x in isa_lint.lint_text(bad)["violations"])
    ok = bad.replace(".LBB0_9:\n", ".LBB0_9:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n")
    assert isa_lint.lint_text(ok)["violations"] == []


def test_unconsumed_load_and_partial_marker_are_refused():
    bad = GOOD.replace("NTLINT_CONSUME v[0:3] v9", "NTLINT_CONSUME v[0:3]")
    assert any("never named" in x for x in isa_lint.lint_text(bad)["violations"])


def test_build_reports_cover_every_site_and_are_clean(built_lib):
    reports = sorted(glob.glob(os.path.join(ROOT, "nthash_amd", "build", "capi_*.o.lint.json")))
    assert len(reports) >= 15, "lint reports of the build are missing (python -m nthash_amd.build --force)"
    loads = markers = kernels = 0
    by_unit = {}
    for f in reports:
        rep = json.load(open(f))
        assert rep["violations"] == [], (os.path.basename(f), rep["violations"][:3])
        by_unit[rep["unit"]] = rep
        loads += rep["hidden_loads"]
        markers += rep["consume_markers"]
        kernels += len(rep["kernels"])
        for name, st in rep["kernels"].items():
            assert st["hidden_loads"] >= 1 and st["consume_markers"] >= 1, (name, st)
    # the units that instantiate the five kernels with such sites (headline, general run-split -- dense, N-aware and
    # its fused consumers --, ragged, dense seeds)
    for unit in ("capi_kmer_runs.hip", "capi_kmer_gen.hip", "capi_kmer_na.hip", "capi_kmer_ragged.hip", "capi_seed.hip",
                 "capi_sink_bloom.hip", "capi_sink_minhash.hip"):
        assert by_unit[unit]["hidden_loads"] > 0, unit
    assert kernels >= 100 and loads >= 400 and markers >= kernels


@pytest.mark.timeout(600)
def test_deliberately_broken_unit_is_refused(tmp_path):
    """-DNT_LINT_SELFTEST=1 reads a register of the prefetched slab between its load and its wait; build.py must refuse
    the unit (here: the general run-split kernel's dense instantiations)"""
    from nthash_amd import build as nb
    hipcc = nb._hipcc()
    rep = nb._lint_unit(hipcc, "capi_kmer_gen.hip", str(tmp_path), ["-DNT_LINT_SELFTEST=1"], False)
    assert rep["hidden_loads"] >= 30
    assert len(rep["violations"]) >= len(rep["kernels"]) >= 10
    assert all("may still be in flight" in v for v in rep["violations"])
    with pytest.raises(RuntimeError, match="ISA lint"):
        nb._compile_unit(hipcc, "capi_kmer_gen.hip", str(tmp_path), ["-DNT_LINT_SELFTEST=1"], True, False)
    assert not os.path.exists(os.path.join(str(tmp_path), "capi_kmer_gen.o"))


def test_build_pins_the_compiler(monkeypatch):
    """round 4: build.py names the hipcc the counted waits and the lint expectations were made with, refuses another one
    unless told that it has been looked at, and every unit's lint report says which compiler produced the ISA it checked"""
    import glob
    import json

    from nthash_amd import build as nb
    assert nb.hipcc_version() == nb.EXPECTED_HIPCC == nb.check_hipcc()
    monkeypatch.setattr(nb, "EXPECTED_HIPCC", "HIP version: 0.0 / AMD clang version none")
    with pytest.raises(RuntimeError, match="review nthash_amd/isa_lint.py"):
        nb.check_hipcc()
    monkeypatch.setenv("NTHASH_AMD_ALLOW_HIPCC_MISMATCH", "1")
    assert nb.check_hipcc() == nb.hipcc_version()
    reports = glob.glob(os.path.join(nb.OBJ, "capi_*.o.lint.json"))
    assert len(reports) >= 17
    stamped = [json.load(open(r)).get("hipcc") for r in reports]
    assert any(stamped) and all(v in (None, nb.hipcc_version()) for v in stamped)   # (None: a unit built before the stamp existed)

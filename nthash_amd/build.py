"""Build the nthash_amd native libraries in-tree with hipcc (gfx950 only).

    python -m nthash_amd.build [--force] [--tag T --flags "-DX=1 ..." [--units capi_kmer_runs,...]]

libnthash_hip.so  the C-ABI (include/nthash_hip.h): HIP kernels + launch logic, one object per
                  csrc/capi_*.hip, compiled in parallel and rebuilt only when a file it includes changed
libnthash.so      the C++ host facade (include/nthash/nthash.hpp) on top of it

--tag T builds a second copy of the C-ABI library with extra compiler flags for in-process A/B timing
(nthash_amd/lib/ab/libnthash_hip_T.so; use it with NTHASH_AMD_LIB=...).
"""
import concurrent.futures
import json
import os
import re
import shlex
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
HIP_SO = os.path.join(LIB, "libnthash_hip.so")
FACADE_SO = os.path.join(LIB, "libnthash.so")
ARCH = "gfx950"


# The compiler the kernels' hidden-load sites were written and linted against (nthash_amd/isa_lint.py: nine sites issue the
# next tile's loads in inline asm and consume them behind a counted s_waitcnt -- what hipcc makes of the code around them
# is checked on the emitted ISA, and the counted waits' argument at each site assumes this code generator).  Another hipcc
# may be fine; it has to be looked at before it builds the product: NTHASH_AMD_ALLOW_HIPCC_MISMATCH=1 after doing so.
EXPECTED_HIPCC = "HIP version: 7.2.26015-fc0010cf6a / AMD clang version 22.0.0git roc-7.2.0 26014"
_hipcc_seen = {}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: nthash_amd needs the ROCm toolchain to build")


def hipcc_version(hipcc=None):
    """'HIP version: X / AMD clang version Y roc-Z N' of the compiler in use (cached)"""
    hipcc = hipcc or _hipcc()
    if hipcc not in _hipcc_seen:
        text = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        hip = re.search(r"HIP version:\s*(\S+)", text)
        clang = re.search(r"clang version (\S+) \(\S+ (roc-\S+) (\d+)", text)
        _hipcc_seen[hipcc] = "HIP version: %s / AMD clang version %s" % (
            hip.group(1) if hip else "?", " ".join(clang.groups()) if clang else "?")
    return _hipcc_seen[hipcc]


def check_hipcc(hipcc=None):
    got = hipcc_version(hipcc)
    if got != EXPECTED_HIPCC and not os.environ.get("NTHASH_AMD_ALLOW_HIPCC_MISMATCH"):
        raise RuntimeError(f"hipcc is '{got}', the kernels' counted waits and the ISA lint expectations were made with "
                           f"'{EXPECTED_HIPCC}': review nthash_amd/isa_lint.py's report, then set NTHASH_AMD_ALLOW_HIPCC_MISMATCH=1")
    return got


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _deps_of(depfile):
    """Files named by a make-style dependency file written by -MD."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    _, _, rhs = text.partition(":")
    return [p for p in rhs.split() if p]


def hip_units():
    return sorted(f for f in os.listdir(CSRC) if f.startswith("capi_") and f.endswith(".hip"))


def _lint_unit(hipcc, unit, objdir, extra, verbose):
    """Device assembly of the unit (hipcc -S --cuda-device-only, the flags of the object) through the ISA lint
    (nthash_amd/isa_lint.py): the hidden-load sites of the kernels must be what the source says they are."""
    from nthash_amd import isa_lint
    src = os.path.join(CSRC, unit)
    asm = os.path.join(objdir, unit[:-4] + ".lint.s")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-command-line-argument",
           f"-I{os.path.join(ROOT, 'include')}"] + list(extra) + ["-S", "--cuda-device-only", src, "-o", asm]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:   # (the object compiled a moment ago: say what the assembly-only compile has to say)
        raise RuntimeError(f"{unit}: the ISA-lint compile failed ({' '.join(cmd)}):\n" + r.stderr[-4000:])
    rep = isa_lint.lint_file(asm)
    os.remove(asm)
    rep["unit"] = unit
    rep["hipcc"] = hipcc_version(hipcc)
    return rep


def _compile_unit(hipcc, unit, objdir, extra, force, verbose):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(objdir, unit[:-4] + ".o")
    dep = obj + ".d"
    lint_json = obj + ".lint.json"
    deps = _deps_of(dep)
    if not force and deps is not None and not _newer(obj, deps + [src]) and all(os.path.exists(d) for d in deps) and \
            os.path.exists(lint_json) and not _newer(lint_json, [obj, os.path.join(HERE, "isa_lint.py")]):
        return obj, False
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed",
           "-Rpass-analysis=kernel-resource-usage",
           f"-I{os.path.join(ROOT, 'include')}", "-MD", "-MF", dep] + list(extra) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    kernels = _resource_usage(r.stderr)
    rest = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and not _REMARK_ECHO.match(l))
    if r.returncode != 0 or "warning:" in rest or "error:" in rest:  # (include chains of the remarks alone are noise)
        print(rest, file=sys.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(obj + ".res.json", "w") as f:
        json.dump(kernels, f, indent=0)
    # No kernel may spill.  Several of them keep loads in flight behind inline asm that hipcc cannot see (the next
    # tile's slab): a register of such a load that is spilled is saved before the load has landed -- wrong hashes,
    # found on seed_wtile_kernel<8> (k = 64 seeds) in round 2.  Scratch traffic would also break the counted waits.
    spilling = [k_ for k_ in kernels if k_.get("scratch", 0) or k_.get("vgpr_spill", 0)]
    if spilling and not os.environ.get("NTHASH_AMD_ALLOW_SPILLS"):
        os.remove(obj)
        raise RuntimeError(f"{unit}: kernels spill registers (scratch bytes per lane): " +
                           ", ".join(f"{k_['name']}={k_.get('scratch', 0)}" for k_ in spilling))
    # ... and what the spill check cannot see: a copy or a use of such a register before its counted wait
    rep = _lint_unit(hipcc, unit, objdir, extra, verbose)
    with open(lint_json, "w") as f:
        json.dump(rep, f, indent=0)
    if rep["violations"] and not os.environ.get("NTHASH_AMD_ALLOW_LINT_FAIL"):
        os.remove(obj)
        raise RuntimeError(f"{unit}: ISA lint of the hidden-load sites failed ({len(rep['violations'])} violations):\n  " +
                           "\n  ".join(rep["violations"][:12]))
    return obj, True


_REMARK_ECHO = re.compile(r"^\s*(\d+ \||\||\^)")  # the source line / caret clang prints under every remark


def _resource_usage(text):
    """[{name, vgprs, sgprs, scratch, vgpr_spill, occupancy, lds}] from -Rpass-analysis=kernel-resource-usage remarks."""
    out, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        key, _, val = m.group(1).partition(":")
        key, val = key.strip(), val.strip()
        if key == "Function Name":
            cur = {"name": val}
            out.append(cur)
        elif cur is not None:
            field = {"VGPRs": "vgprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
                     "VGPRs Spill": "vgpr_spill", "Occupancy [waves/SIMD]": "occupancy",
                     "LDS Size [bytes/block]": "lds"}.get(key)
            if field and val.lstrip("-").isdigit():
                cur[field] = int(val)
    return out


def build_hip(out_so=HIP_SO, objdir=OBJ, extra=(), force=False, verbose=False, only_units=None):
    """only_units: compile just these units with `extra` into objdir and take every other object from the
    base build (a variant of one kernel does not need the other eleven units rebuilt)."""
    os.makedirs(os.path.dirname(out_so), exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    check_hipcc(hipcc)
    units = hip_units()
    mine = [u for u in units if only_units is None or u[:-4] in only_units]
    jobs = min(len(mine), max(1, (os.cpu_count() or 2)))
    with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        res = list(pool.map(lambda u: _compile_unit(hipcc, u, objdir, extra, force, verbose), mine))
    if only_units is not None:
        build_hip(verbose=verbose)  # the base objects
        res += [(os.path.join(OBJ, u[:-4] + ".o"), False) for u in units if u not in mine]
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or _newer(out_so, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", out_so, "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out_so


def build(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    build_hip(force=force, verbose=verbose)
    # the C++ host facade is plain host C++17 on top of the C-ABI
    facade_src = os.path.join(CSRC, "nthash_facade.cpp")
    deps = [facade_src, os.path.join(CSRC, "nt_math.hpp"), os.path.join(CSRC, "seed_parse.hpp"),
            os.path.join(ROOT, "include", "nthash", "nthash.hpp"),
            os.path.join(ROOT, "include", "nthash_hip.h"), HIP_SO]
    if force or _newer(FACADE_SO, deps):
        cxx = shutil.which("g++") or shutil.which("c++") or _hipcc()
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-pthread",
               f"-I{os.path.join(ROOT, 'include')}", facade_src, "-o", FACADE_SO,
               f"-L{LIB}", "-lnthash_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HIP_SO


def build_variant(tag, flags, force=False, verbose=False, only_units=None):
    out = os.path.join(LIB, "ab", f"libnthash_hip_{tag}.so")
    return build_hip(out_so=out, objdir=os.path.join(OBJ, "ab_" + tag), extra=flags, force=force, verbose=verbose,
                     only_units=only_units)


if __name__ == "__main__":
    argv = sys.argv[1:]
    force = "--force" in argv
    if "--tag" in argv:
        tag = argv[argv.index("--tag") + 1]
        flags = shlex.split(argv[argv.index("--flags") + 1]) if "--flags" in argv else []
        units = argv[argv.index("--units") + 1].split(",") if "--units" in argv else None
        print(build_variant(tag, flags, force=force, verbose=True, only_units=units))
    else:
        build(force=force, verbose=True)
        print(HIP_SO)

"""ASan / UBSan job (the reference is sanitizer-clean) for the host code that has hand-written
cursor, window and position logic: the C++ facade (nthash_amd/csrc/nthash_facade.cpp with nt_math.hpp and
seed_parse.hpp) and the C restatement (oracle/nthash_oracle.c).  Everything is compiled with
-fsanitize=address,undefined -fno-sanitize-recover=all and a driver walks NtHash / BlindNtHash / SeedNtHash /
BlindSeedNtHash through random sequences and call sequences, comparing every answer with the oracle.

No GPU here: the facade links against a STUB of the C-ABI (tests/sanitize/capi_stub.c, test infrastructure only) that
grants a context and refuses to hash, so what runs is exactly the facade's host side -- the recurrences behind
roll_back()/peek()/Blind* and behind roll() on short sequences, the skipping state machines, copies and moves."""
import os
import subprocess

from conftest import ROOT


def test_facade_and_oracle_under_asan_ubsan(tmp_path):
    inc = os.path.join(ROOT, "include")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    stub = tmp_path / "libnthash_hip.so"
    subprocess.check_call(["gcc", "-std=c11", "-fPIC", "-shared", f"-I{inc}", os.path.join(ROOT, "tests", "sanitize", "capi_stub.c"),
                           "-o", str(stub)])
    oracle_o = tmp_path / "oracle.o"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", *san, "-c", os.path.join(ROOT, "oracle", "nthash_oracle.c"),
                           "-o", str(oracle_o)])
    exe = tmp_path / "facade_sanitize"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", *san, f"-I{inc}",
                           os.path.join(ROOT, "tests", "sanitize", "facade_sanitize_driver.cpp"),
                           os.path.join(ROOT, "nthash_amd", "csrc", "nthash_facade.cpp"), str(oracle_o),
                           f"-L{tmp_path}", "-lnthash_hip", f"-Wl,-rpath,{tmp_path}", "-pthread", "-o", str(exe)])
    env = dict(os.environ)
    env.update(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               NTHASH_AMD_FORCE_DEVICE="0")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "sanitize driver OK" in r.stdout

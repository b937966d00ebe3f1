#!/usr/bin/env python3
"""Whole-stream checksums of bench.py's full-size workloads, from the REAL reference (oracle/_ref, OpenMP over
reads; run in the build container where /root/reference exists):

    python tests/golden/gen_bench_checksums.py            # ~10 minutes on 8 cores

Writes tests/golden/bench_checksums.json: for every (workload, first_read, n_reads) the wrapping sum and XOR of
every hash the reference emits -- what bench.py compares the device's on-device checksum of the full stream with
(SURVEY 8d "Correctness at scale").  Data only: inputs are the counter-based synthetic reads (seed 42).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Reference  # noqa: E402

SEED_A = "1010101010101010101010101010101"
SEED_B = "1101101101101101011011011011011"
OUT = os.path.join(ROOT, "tests", "golden", "bench_checksums.json")

JOBS = [("c2", 0, 100_000_000, 150, 31, 1, None)]
JOBS += [("c2", r * 125_000_000, 125_000_000, 150, 31, 1, None) for r in range(8)]  # BASELINE config 5 shards
JOBS += [("c3", 0, 100_000_000, 150, 31, 4, None),
         ("c4", 0, 50_000_000, 250, 31, 3, [SEED_A, SEED_B]),
         ("ref", 0, 100_000_000, 100, 64, 3, None)]
# small entries of the same form, so that the CPU tests can pin this file's format against the oracle
JOBS += [("c2", 0, 20_000, 150, 31, 1, None), ("c2", 125_000_000, 20_000, 150, 31, 1, None),
         ("c4", 0, 5_000, 250, 31, 3, [SEED_A, SEED_B])]


# variable-length reads (bench.py's "var" line): (name, first_read, n_reads, len_min, len_max, k, m)
VAR_JOBS = [("var", 0, 20_000_000, 100, 150, 31, 1), ("var", 0, 20_000, 100, 150, 31, 1)]
# the same rule with one length (bench.py's "c2_dirty" lines: fixed-length reads, an N in one read of ~1000)
VAR_JOBS += [("c2_dirty", 0, 20_000_000, 150, 150, 31, 1), ("c2_dirty", 0, 20_000, 150, 150, 31, 1)]


def main():
    ref = Reference()
    assert ref.has_synth
    try:
        done = {(e["workload"], e["first_read"], e["n_reads"]): e for e in json.load(open(OUT))}
    except (OSError, ValueError):
        done = {}
    for (name, first, n, L, k, m, seeds) in JOBS:
        if (name, first, n) in done:
            continue
        t0 = time.time()
        s, x, tot = ref.synth_checksum(first, n, L, k, m, seeds=seeds)
        done[(name, first, n)] = {"workload": name, "first_read": first, "n_reads": n, "len": L, "k": k, "m": m,
                                  "seeds": seeds, "seed": 42, "total": tot, "sum": format(s, "016x"),
                                  "xor": format(x, "016x"), "source": ref.fn_name() + " (oracle/_ref)"}
        print(name, first, n, format(s, "016x"), format(x, "016x"), tot, "%.1fs" % (time.time() - t0), flush=True)
        json.dump(sorted(done.values(), key=lambda e: (e["workload"], e["n_reads"], e["first_read"])),
                  open(OUT, "w"), indent=0)


    for (name, first, n, lmin, lmax, k, m) in VAR_JOBS:
        if (name, first, n) in done:
            continue
        t0 = time.time()
        s, x, tot = ref.synth_var_checksum(first, n, lmin, lmax, k, m)
        done[(name, first, n)] = {"workload": name, "first_read": first, "n_reads": n, "len_min": lmin, "len": lmax, "k": k,
                                  "m": m, "seeds": None, "seed": 42, "total": tot, "sum": format(s, "016x"),
                                  "xor": format(x, "016x"), "source": ref.fn_name() + " (oracle/_ref)"}
        print(name, first, n, format(s, "016x"), format(x, "016x"), tot, "%.1fs" % (time.time() - t0), flush=True)
        json.dump(sorted(done.values(), key=lambda e: (e["workload"], e["n_reads"], e["first_read"])),
                  open(OUT, "w"), indent=0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libnthash_ref.so).

Run in the build container (needs /root/reference to have been compiled by
oracle/Makefile):   python tests/golden/gen_golden.py

The fixtures are data only: inputs (sequences, k, m, seeds, API call scripts)
and the outputs the reference produced for them.  They travel to the GPU box;
the reference itself does not.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Reference, concat_reads  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED_A = "1010101010101010101010101010101"
SEED_B = "1101101101101101011011011011011"


def hx(a):
    return [format(int(x), "016x") for x in np.asarray(a).ravel()]


def main():
    ref = Reference()
    rng = np.random.default_rng(20260928)
    alph_clean = np.frombuffer(b"ACGT", dtype=np.uint8)
    alph_mixed = np.frombuffer(b"ACGTacgtUuNnRYKMSW-*X", dtype=np.uint8)

    def rand_read(L, mode):
        if mode == "clean":
            return alph_clean[rng.integers(0, 4, L)].tobytes().decode()
        if mode == "mixed":  # mostly bases, some non-bases
            idx = np.where(rng.random(L) < 0.93, rng.integers(0, 10, L), rng.integers(10, len(alph_mixed), L))
            return alph_mixed[idx].tobytes().decode()
        idx = rng.integers(0, len(alph_mixed), L)
        return alph_mixed[idx].tobytes().decode()

    # ---- 1. k-mer batches ----------------------------------------------------
    kmer_cases = []
    fixed = [
        (["ACATGCATGCA"], 5, 3),                       # reference tests.cpp:50-57
        (["AGTCAGTC"], 4, 3),
        (["ACGTACACTGGACTGAGTCT"], 18, 3),
        (["ACGTACACTGNNCTGAGTCT"], 8, 3),              # skipping Ns, tests.cpp:181-208
        (["ACGUACACUGGACUGAGUCUACGG"], 20, 3),         # RNA
        (["CACTCGGCCACACACACACACACACACCCTCACACACACAAAACGCACAC"], 31, 4),
        (["ACGTAC"], 3, 2), (["ACGTAC"], 4, 2), (["ACGTAC"], 5, 2), (["ACGTAC"], 6, 2),
        (["ATCGTACGATGCATGCATGCTGACG"], 6, 3),         # examples/kmer_hashing.cpp
        (["ACGTNACGTACGNTACGTACG"], 4, 1),
        (["NNNNNNNN", "ACGT", "AC", "", "acgtacgtnACGT"], 4, 2),
    ]
    for reads, k, m in fixed:
        kmer_cases.append((reads, k, m))
    for _ in range(40):
        n = int(rng.integers(1, 5))
        k = int(rng.choice([3, 4, 7, 15, 16, 17, 21, 31, 32, 33, 47, 64, 70]))
        m = int(rng.integers(1, 6))
        mode = str(rng.choice(["clean", "mixed", "dirty"]))
        reads = [rand_read(int(rng.integers(0, 160)), mode) for _ in range(n)]
        kmer_cases.append((reads, k, m))
    out = []
    for reads, k, m in kmer_cases:
        d, offs = concat_reads(reads)
        r = ref.kmer_batch(d, offs, k, m, want_strands=True)
        out.append({"reads": reads, "k": k, "m": m, "counts": [int(x) for x in r["counts"]],
                    "pos": [int(x) for x in r["pos"]], "hashes": hx(r["hashes"]),
                    "fwd": hx(r["fwd"]), "rev": hx(r["rev"])})
    json.dump(out, open(os.path.join(OUT, "kmer_cases.json"), "w"), indent=0)

    # ---- 2. spaced-seed batches ---------------------------------------------
    seed_cases = [
        (["ACATGCATGCA"], ["11100111"], 3),            # reference tests.cpp:231-240
        (["ACGTACACTGGACTGAGTCT"], ["111110000000011111", "111111100001111111"], 2),
        (["ACTAGCTG"], ["110011"], 3),
        (["CACTCGGCCACACACACACACACACACCCTCACACACACAAAACGCACAC"], [SEED_A, SEED_B], 3),
        (["ACGTTGCATGCATGCAAACCCGGGTTTACGATCGATCGTAGC"], ["1111111111111110111111111111111"], 2),
        (["ACNTACGTACGTAC"], ["110011"], 2),
        (["ACGTACGTNCGTACGTAC"], ["110011"], 2),
        (["ATGCTAGTAGCTGAC"], ["110011", "101101"], 3),
        (["ATGCTAGTAGCTGAC"], ["11111"], 3),
        (["CACTCGGCCACACACACACACACACACCCTCACACACACAAAACGCACAC"],
         ["11011000001100101101011000011010110100110000011011",
          "00000000000000000000000011000000000000000000000000",
          "11111111111111111111111100111111111111111111111111",
          "11111111111111111111111111111111111111111111111111"], 4),
    ]
    for _ in range(40):
        n = int(rng.integers(1, 4))
        k = int(rng.choice([3, 5, 8, 16, 17, 21, 31, 32, 33, 48, 50, 64, 70]))
        m2 = int(rng.integers(1, 5))
        mode = str(rng.choice(["clean", "mixed", "dirty"]))
        reads = [rand_read(int(rng.integers(0, 170)), mode) for _ in range(n)]
        seeds = []
        for _s in range(int(rng.integers(1, 4))):
            dens = rng.random()
            half = "".join("1" if rng.random() < dens else "0" for _ in range((k + 1) // 2))
            s = half + half[: k // 2][::-1]  # palindromic (no warning noise)
            if rng.random() < 0.25:  # asymmetric on purpose
                s = "".join("1" if rng.random() < dens else "0" for _ in range(k))
            seeds.append(s)
        seed_cases.append((reads, seeds, m2))
    out = []
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # silence the reference's asymmetric-seed warnings
    try:
        for reads, seeds, m2 in seed_cases:
            k = len(seeds[0])
            d, offs = concat_reads(reads)
            r = ref.seed_batch(d, offs, seeds, k, m2)
            out.append({"reads": reads, "seeds": seeds, "k": k, "m2": m2,
                        "counts": [int(x) for x in r["counts"]], "pos": [int(x) for x in r["pos"]],
                        "hashes": hx(r["hashes"])})
    finally:
        os.dup2(saved, 2)
    json.dump(out, open(os.path.join(OUT, "seed_cases.json"), "w"), indent=0)

    # ---- 3. synthetic-read checksums (BASELINE configs, counter-based reads) --
    synth = []
    for (n, L, k, m) in [(10000, 150, 31, 1), (10000, 150, 31, 4), (2000, 100, 64, 3), (3000, 151, 21, 2)]:
        data = ref.synth_reads(0, n, L, 42)
        offs = np.arange(n + 1, dtype=np.uint64) * L
        r = ref.kmer_batch(data, offs, k, m, want_pos=False)
        s, x = ref.checksum(r["hashes"])
        synth.append({"kind": "kmer", "n_reads": n, "len": L, "k": k, "m": m, "seed": 42,
                      "total": int(r["total"]), "sum": format(s, "016x"), "xor": format(x, "016x"),
                      "first_read": data[:L].tobytes().decode(), "head": hx(r["hashes"][:4])})
    for (n, L, m2) in [(4000, 250, 3), (1000, 150, 2)]:
        data = ref.synth_reads(0, n, L, 42)
        offs = np.arange(n + 1, dtype=np.uint64) * L
        r = ref.seed_batch(data, offs, [SEED_A, SEED_B], 31, m2, want_pos=False)
        s, x = ref.checksum(r["hashes"])
        synth.append({"kind": "seed", "n_reads": n, "len": L, "k": 31, "m2": m2, "seeds": [SEED_A, SEED_B],
                      "seed": 42, "total": int(r["total"]), "sum": format(s, "016x"),
                      "xor": format(x, "016x"), "head": hx(r["hashes"][:2])})
    json.dump(synth, open(os.path.join(OUT, "synth_checksums.json"), "w"), indent=0)

    # ---- 4. iterator API scripts (host facade parity) -----------------------
    scripts = []

    def nth(seq, m, k, pos0, ops):
        res = ref.nthash_script(seq, m, k, pos0, ops)
        scripts.append({"cls": "NtHash", "seq": seq, "m": m, "k": k, "pos0": pos0, "ops": ops,
                        "ret": [a[0] for a in res], "pos": [a[1] for a in res],
                        "fwd": [format(a[2], "016x") for a in res],
                        "rev": [format(a[3], "016x") for a in res],
                        # hashes() is uninitialised heap memory until a call succeeds: only
                        # record it for calls that returned true
                        "hashes": [hx(a[4]) if a[0] else None for a in res]})

    nth("ACATGCATGCA", 3, 5, 0, "rrrrrrrrr")
    nth("ACTAGCTG", 3, 5, 0, "rrrrrbbbbb")                 # tests.cpp:135-157
    nth("ACTGATCAG", 3, 6, 0, "rpPCrpPArpPGr")              # peeking
    nth("ACGTACACTGNNCTGAGTCT", 3, 8, 0, "r" * 12 + "b" * 6 + "r" * 4)
    nth("ACGTACACTGNNCTGAGTCT", 2, 4, 3, "rrqQAqQNbbpPNrr")
    nth("NNACGTACGTNN", 2, 4, 0, "rrrrrrrbbbbbb")
    nth("ACGTACGTAC", 1, 10, 0, "rrbqp")
    nth("ACGTACGTACGTTTGACCA", 4, 7, 5, "pqrrbbbbbbrr")
    for _ in range(12):
        L = int(rng.integers(12, 60))
        k = int(rng.integers(3, min(L, 33)))
        seq = rand_read(L, str(rng.choice(["clean", "mixed"])))
        ops = "".join(rng.choice(list("rrrrbpq")) for _ in range(40))
        ops = "".join(o + ("ACGTN"[int(rng.integers(0, 5))] if o in "PQ" else "") for o in ops)
        nth(seq, int(rng.integers(1, 4)), k, int(rng.integers(0, L - k + 1)), ops)

    def blind(seq, m, k, pos0, ops):
        res = ref.blind_script(seq, m, k, pos0, ops)
        scripts.append({"cls": "BlindNtHash", "seq": seq, "m": m, "k": k, "pos0": pos0, "ops": ops,
                        "pos": [a[0] for a in res], "fwd": [format(a[1], "016x") for a in res],
                        "rev": [format(a[2], "016x") for a in res], "hashes": [hx(a[3]) for a in res]})

    blind("ACATGCATGCA", 3, 5, 0, "RGRCRARTPGQTBABC")
    blind("ACATGCATGCA", 3, 5, 2, "RARCPT")
    blind("ACGTACGTACGTACGTACGTACGTACGTACGTACG", 2, 31, 0, "RARCRGRTRNBABCQGPT")

    def seed(seq, seeds, m2, pos0, ops):
        k = len(seeds[0])
        res = ref.seed_script(seq, seeds, m2, k, pos0, ops)
        scripts.append({"cls": "SeedNtHash", "seq": seq, "seeds": seeds, "m2": m2, "k": k, "pos0": pos0,
                        "ops": ops, "ret": [a[0] for a in res], "pos": [a[1] for a in res],
                        "fwd": [hx(a[2]) if a[0] else None for a in res],
                        "rev": [hx(a[3]) if a[0] else None for a in res],
                        "hashes": [hx(a[4]) if a[0] else None for a in res]})

    seed("ACATGCATGCA", ["11100111"], 3, 0, "rrrr")
    seed("ACTAGCTG", ["110011"], 3, 0, "rrrbbb")           # tests.cpp:324-347
    seed("ACGTACACTGGACTGAGTCT", ["111110000000011111", "111111100001111111"], 2, 0, "rrqQArpPGrb")
    seed("ACNTACGTACGTAC", ["110011"], 2, 0, "rrrrrrrrrbbb")
    seed("ATGCTAGTAGCTGAC", ["110011", "101101"], 3, 0, "rrrrpqbbrr")
    seed("ACGTTGCATGCATGCAAACCCGGGTTTACGATCGATCGTAGC", ["1111111111111110111111111111111"], 2, 0,
         "rrrrbbpqrr")

    def blindseed(seq, seeds, m2, pos0, ops):
        k = len(seeds[0])
        res = ref.blindseed_script(seq, seeds, m2, k, pos0, ops)
        scripts.append({"cls": "BlindSeedNtHash", "seq": seq, "seeds": seeds, "m2": m2, "k": k,
                        "pos0": pos0, "ops": ops, "pos": [a[0] for a in res],
                        "fwd": [hx(a[1]) for a in res], "rev": [hx(a[2]) for a in res],
                        "hashes": [hx(a[3]) for a in res]})

    blindseed("ATGCTAGTAGCTGAC", ["110011", "101101"], 3, 0, "RGRTRARGBABT")
    blindseed("ACCAGT", ["110011", "101101"], 3, 0, "RABA")
    blindseed("ATGCTAGTAGCTGAC", ["110011", "101101"], 1, 3, "RARCRGRT")
    json.dump(scripts, open(os.path.join(OUT, "api_scripts.json"), "w"), indent=0)

    # ---- 5. parse_seeds / get_blocks probes --------------------------------
    ps = []
    for s in [SEED_A, SEED_B, "11100111", "101101", "110011", "1111", "0110", "1x01", "11111111111111101111111111"]:
        ps.append({"seed": s, "dont_care": ref.parse_seeds(s)})
    json.dump(ps, open(os.path.join(OUT, "parse_seeds.json"), "w"), indent=0)
    # ---- 6. batched graph-extension query: BlindNtHash::peek / peek_back per base -------
    ext = []
    for _ in range(40):
        k = int(rng.choice([3, 5, 16, 21, 31, 32, 33, 64, 100]))
        m = int(rng.integers(1, 5))
        # bases only (any case, U allowed): for other bytes the reference's constructor indexes its
        # tetramer tables with CONVERT_TAB = 255 (src/kmer.cpp:50-54), which is not a defined hash
        kmer = "".join("ACGTacgtUu"[i] for i in rng.integers(0, 10, k))
        res = ref.blind_script(kmer, m, k, 0, "PAPCPGPTQAQCQGQT")
        ext.append({"kmer": kmer, "k": k, "m": m, "self": hx(res[0][3]),
                    "next": [hx(res[1 + b][3]) for b in range(4)],
                    "prev": [hx(res[5 + b][3]) for b in range(4)]})
    json.dump(ext, open(os.path.join(OUT, "extend_cases.json"), "w"), indent=0)
    # ---- 7. the same query through spaced seeds: BlindSeedNtHash::roll(c) / roll_back(c) per base, each from a fresh object
    # (src/seed.cpp:701-737).  Seeds with and without monomers (roll_back reads a monomer's base from the window it
    # leaves, src/seed.cpp:195-198), the don't-care description (get_blocks' second branch), asymmetric seeds, k to 100
    rng7 = np.random.default_rng(7)
    sext = []

    def rand_seed(k, kind):
        if kind == "blocks":       # runs of at least two
            s, p = "", 0
            while len(s) < k:
                run = int(rng7.integers(2, 7))
                s += ("1" if p % 2 == 0 else "0") * run
                p += 1
            s = s[:k]
            return s if not (s[-1] != s[-2]) else s[:-1] + s[-2]
        if kind == "dense":        # mostly care: the don't-care description wins
            a = ["1"] * k
            for i in rng7.choice(k, max(1, k // 12), replace=False):
                a[int(i)] = "0"
            return "".join(a)
        return "".join("10"[int(x)] for x in rng7.integers(0, 2, k))   # anything: monomers everywhere
    for i in range(48):
        k = int(rng7.choice([6, 11, 16, 21, 31, 32, 33, 48, 64, 100]))
        n_seeds = int(rng7.integers(1, 4))
        seeds7 = [rand_seed(k, ["blocks", "dense", "any"][(i + j) % 3]) for j in range(n_seeds)]
        m2 = int(rng7.integers(1, 4))
        kmer = "".join("ACGTacgtUu"[j] for j in rng7.integers(0, 10, k))
        base = ref.blindseed_script(kmer, seeds7, m2, k, 0, "")
        nxt = [hx(ref.blindseed_script(kmer, seeds7, m2, k, 0, "R" + b)[1][3]) for b in "ACGT"]
        prv = [hx(ref.blindseed_script(kmer, seeds7, m2, k, 0, "B" + b)[1][3]) for b in "ACGT"]
        sext.append({"kmer": kmer, "k": k, "seeds": seeds7, "m2": m2, "self": hx(base[0][3]), "next": nxt, "prev": prv})
    json.dump(sext, open(os.path.join(OUT, "seed_extend_cases.json"), "w"), indent=0)
    print("fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()

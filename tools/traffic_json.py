#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the PMC summaries of an evidence round (tools/run_pmc.sh: separate --pmc passes of
FETCH_SIZE and WRITE_SIZE over the headline kernel and over a 4 GiB copy that calibrates them).

    python tools/traffic_json.py r05
HBM bytes per k-mer = (FETCH_SIZE x 1024 x the copy's correction + WRITE_SIZE x 1024) / k-mers of the launch; the correction is
what makes the copy's FETCH_SIZE equal the bytes it read (2 on gfx950: /opt/skills/guides/MI355X_MICROARCH.md)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
SHAPES = {"c2": dict(file=f"{tag}_pmc_c2_summary.txt", kernel="kmer_runs", reads=20_000_000, L=150, k=31, per=1, alg=9.25),
          "c4": dict(file=f"{tag}_pmc_c4_summary.txt", kernel="seed_wtile", reads=8_000_000, L=250, k=31, per=6, alg=(250 + 220 * 48) / 220)}
COPY_BYTES = 4 << 30


def rows(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+n=\s*(\d+)\s+avg=(\S+)", line)
        if m:
            out[(m.group(1), m.group(2))] = float(m.group(4))
    return out


res = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/run_pmc.sh via tools/evidence_round.sh {tag}; "
                 f"made by tools/traffic_json.py from profiles/{tag}_pmc_*_summary.txt"}
for name, sh in SHAPES.items():
    path = os.path.join(ROOT, "profiles", sh["file"])
    if not os.path.exists(path):
        continue
    r = rows(path)
    corr = COPY_BYTES / (r[("copy", "FETCH_SIZE")] * 1024.0)
    if "calibration" not in res:
        res["calibration"] = {"copy_bytes": COPY_BYTES, "FETCH_SIZE_kb": r[("copy", "FETCH_SIZE")], "WRITE_SIZE_kb": r[("copy", "WRITE_SIZE")],
                              "fetch_correction": corr}
    kern = [k for k in r if k[0].startswith(sh["kernel"]) and k[1] == "FETCH_SIZE"]
    if not kern:
        continue
    kn = kern[0][0]
    kmers = sh["reads"] * (sh["L"] - sh["k"] + 1)
    rd, wr = r[(kn, "FETCH_SIZE")] * 1024.0 * round(corr), r[(kn, "WRITE_SIZE")] * 1024.0
    res[name] = {"reads": sh["reads"], "kmers": kmers, "FETCH_SIZE_kb": r[(kn, "FETCH_SIZE")], "WRITE_SIZE_kb": r[(kn, "WRITE_SIZE")],
                 "hbm_read_bytes": rd, "hbm_write_bytes": wr, "bytes_per_kmer_algorithmic": sh["alg"],
                 "bytes_per_kmer_measured": (rd + wr) / kmers}
out = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
json.dump(res, open(out, "w"), indent=1)
print(open(out).read())

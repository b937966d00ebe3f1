#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc CSV output into per-kernel, per-counter averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        short = "mz_fused" if "minimizer_fused" in name else "mz_w" if "minimizer_w_kernel" in name else "copy" if "copy_kernel" in name else "ragged_hash" if "kmer_ragged_kernel<2" in name else \
            "ragged_count" if "kmer_ragged_kernel<1" in name else "reads_hash" if "kmer_reads_kernel<2" in name else \
            "reads_mark" if "kmer_reads_kernel<1" in name else "reads_dirty" if "kmer_dirty_reads" in name else "kmer_runs_gen" if "kmer_runs_gen" in name else \
            "kmer_runs" if "kmer_runs" in name else "kmer_fixed" if "kmer_fixed" in name else \
            "seed_fixed" if "seed_fixed" in name else "seed_wtile" if "seed_wtile" in name else None
        if short is None:
            continue
        acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (kern, ctr), vals in sorted(acc.items()):
    print(f"{kern:12s} {ctr:28s} n={len(vals):2d} avg={sum(vals)/len(vals):.6g} last={vals[-1]:.6g}")

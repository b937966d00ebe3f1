#!/usr/bin/env python3
"""Per-kernel statistics of the FULL-SIZE launches in a rocprofv3 kernel trace.

    python tools/kernel_stats_filter.py <dir with *kernel_trace.csv> <out.csv>

rocprofv3 --stats averages every launch of a kernel: the first-batch autotune trials (slices of a quarter millisecond) and
the spot checks after the timed steps pull that average far below the duration of a timed launch (round 3's `ref` csv:
avg 11.3 ms, max 17.8 ms).  Here a kernel's launches are kept when they last at least half as long as its longest one, the
first of them (cold) is dropped when three or more remain, and what is left is summarised: a fraction recomputed from
avg_ms of this file is the fraction of a timed launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
by = defaultdict(list)
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for name, ls in by.items():
    ls.sort()
    longest = max(d for _, d in ls)
    full = [d for _, d in ls if d >= longest / 2]
    if len(full) >= 3:
        full = full[1:]
    rows.append((sum(full), name, len(ls), len(full), sum(full) / len(full) / 1e6, min(full) / 1e6, max(full) / 1e6))
rows.sort(reverse=True)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_all", "launches_full_size", "avg_ms", "min_ms", "max_ms"])
    for _, name, n_all, n_full, avg, mn, mx in rows:
        w.writerow([name, n_all, n_full, f"{avg:.4f}", f"{mn:.4f}", f"{mx:.4f}"])
print(open(out).read()[:1500])

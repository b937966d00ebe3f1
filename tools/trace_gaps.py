#!/usr/bin/env python3
"""Timeline of the last N kernels of a rocprofv3 --kernel-trace csv: start offset, duration, gap to the kernel before.

    python tools/trace_gaps.py <dir with *kernel_trace.csv> [last=24]
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-last:]
t0, prev_end = int(rows[0]["Start_Timestamp"]), None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:9.1f} us"
    print(f"{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  {gap:20s} {r['Kernel_Name'][:70]}")
    prev_end = e

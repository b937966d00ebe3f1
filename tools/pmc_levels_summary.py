#!/usr/bin/env python3
"""Counters of the Bloom consumers' kernels from a tools/pmc_any.sh directory, normalised by each kernel's duration:
VALU busy, LDS active, share of bank conflicts, time a wave waits on an instruction.

    bash tools/pmc_any.sh gpurun_out/pmc_seed_insert bloom_ python tools/seed_insert_one.py
    python tools/pmc_levels_summary.py gpurun_out/pmc_seed_insert > profiles/rNN_pmc_seed_insert_summary.txt
"""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "bloom_" in name and "overflow" not in name:
            acc[(name.split("(")[0].replace("void ntamd::", "").replace("ntamd::", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
print("# tools/pmc_any.sh <dir> bloom_ python tools/seed_insert_one.py  (config 4's seed pair, 3 hashes per seed, 5 M x 250 bp, 4 GiB filter;")
print("# 3 calls x 2 rounds: the average of the 6 launches of each kernel; SQ_* ACTIVE / WAVE_CYCLES in quad-cycles, see r06_notes.md 1.3;")
print("# a kernel's duration in clocks = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs); 1024 SIMDs, 256 CUs)")
for k in sorted({k for k, _ in acc}):
    g = lambda c: sum(acc[(k, c)]) / max(1, len(acc[(k, c)])) if (k, c) in acc else 0.0
    print(k)
    for c in ("GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
              "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        print(f"    {c:24s} {g(c):.5g}")
    clk = g("GRBM_GUI_ACTIVE") / 8
    if clk and g("SQ_LDS_IDX_ACTIVE") and g("SQ_WAVE_CYCLES"):
        print(f"    -> VALU busy {g('SQ_ACTIVE_INST_VALU') * 4 / (clk * 1024):.2f} of the SIMD time; LDS active {g('SQ_LDS_IDX_ACTIVE') / (clk * 256):.2f} of the CU time, "
              f"{g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.2f} of it bank conflicts; a wave waits on an instruction "
              f"{g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f} of its time")

#!/bin/bash
# SQ counters of every kernel of a command (one counter group per rocprofv3 run): tools/pmc_any.sh <outdir> <kernel-substring> <command ...>
OUT=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
GROUPS_SQ=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
  "GRBM_GUI_ACTIVE GRBM_COUNT")
i=0
for grp in "${GROUPS_SQ[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- "$@" > "$OUT/p$i.log" 2>&1
done
python - "$OUT" "$PAT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if pat in name:
            acc[(name[:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (kern, ctr), vals in sorted(acc.items()):
    print(f"{kern:60s} {ctr:24s} n={len(vals):2d} avg={sum(vals)/len(vals):.6g}")
PY

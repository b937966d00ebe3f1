#!/bin/bash
# PMC passes (one counter group per run, as gpurun requires: --pmc only with --kernel-trace)
# usage: [PMC_SQ_ONLY=1] tools/run_pmc.sh <outdir> <cfg> <reads>
set -u
OUT=${1:-gpurun_out/pmc}; CFG=${2:-c2}; READS=${3:-20000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
GROUPS_SQ=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
  "GRBM_GUI_ACTIVE GRBM_COUNT")
GROUPS_MEM=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum")
if [ -n "${PMC_SQ_ONLY:-}" ]; then ALL=("${GROUPS_SQ[@]}"); else ALL=("${GROUPS_MEM[@]}" "${GROUPS_SQ[@]}"); fi
i=0
for grp in "${ALL[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python tools/pmc_probe.py $CFG $READS > "$OUT/p$i.log" 2>&1
  echo "== pass $i: $grp rc=$?"
done
python tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"

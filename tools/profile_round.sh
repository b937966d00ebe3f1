#!/bin/bash
# Round profile: bench line + rocprofv3 kernel-trace stats of the same command + PMC passes.
# usage (on the GPU box): bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
tail -c 2500 "$OUT/bench_c2.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_c2_under_rocprof.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name "*stats*.csv" | head
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
cat "$OUT/kernel_stats.csv" 2>/dev/null | head -8
# (c3 / c4 / ref ride in the "secondary" object of the default line since round 2)
bash tools/run_pmc.sh "$OUT/pmc_c2" c2 20000000 > "$OUT/pmc_c2.log" 2>&1
cat "$OUT/pmc_c2/summary.txt" | grep -v "^copy"
bash tools/run_pmc.sh "$OUT/pmc_c4" c4 8000000 > "$OUT/pmc_c4.log" 2>&1
cat "$OUT/pmc_c4/summary.txt" | grep -v "^copy"
find "$OUT/pmc_c4" -name "*kernel_trace.csv" -delete
PMC_SQ_ONLY=1 bash tools/run_pmc.sh "$OUT/pmc_rag" rag 10000000 > "$OUT/pmc_rag.log" 2>&1
cat "$OUT/pmc_rag/summary.txt" | grep -v "^copy"
find "$OUT/pmc_rag" -name "*kernel_trace.csv" -delete
# keep only small summaries in the merge-back
find "$OUT" -name "*.db" -delete; find "$OUT/pmc_c2" -name "*kernel_trace.csv" -delete
du -sh "$OUT"

#!/bin/bash
# Round profile: bench line + ONE rocprofv3 --kernel-trace --stats process PER CONFIG (equal launches inside a config, so
# that the kernel's average duration in its csv is the duration of one launch of known size), then the PMC passes.
# usage (on the GPU box): bash tools/profile_round.sh r03
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 1500 "$OUT/bench_default.json"; echo
# the headline config under rocprofv3: the driver's own command
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > "$OUT/bench_c2_under_rocprof.json" 2> "$OUT/trace_c2.err"
# the secondary configs, each alone, launches of one size (chunk-reads divides the reads)
for spec in "c3:50000000" "c4:12500000" "ref:0" "var:0" "var_slots:0" "c2_packed:0"; do
  cfg=${spec%%:*}; chunk=${spec##*:}
  extra=""; if [ "$chunk" != "0" ]; then extra="--chunk-reads $chunk"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -o kt -- python bench.py --config $cfg --steps 5 --warmup 2 $extra --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > "$OUT/bench_${cfg}_under_rocprof.json" 2> "$OUT/trace_$cfg.err"
done
for cfg in c2 c3 c4 ref var var_slots c2_packed; do
  for f in $(find "$OUT/trace_$cfg" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_$cfg.csv"; done
  echo "== $cfg"; head -4 "$OUT/kernel_stats_$cfg.csv" 2>/dev/null
done
# a consumer: per-read minimizers (w = 10) of 20 M clean reads, of the same reads with an N here and there, and given by offsets
for shape in clean dirty var; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mz_$shape" -o kt -- python tools/minimizer_bench.py 20000000 10 $shape > "$OUT/minimizer_${shape}_under_rocprof.txt" 2> "$OUT/trace_mz_$shape.err"
  for f in $(find "$OUT/trace_mz_$shape" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_minimizers_$shape.csv"; done
  cat "$OUT/minimizer_${shape}_under_rocprof.txt"; head -6 "$OUT/kernel_stats_minimizers_$shape.csv" 2>/dev/null
done
bash tools/run_pmc.sh "$OUT/pmc_c2" c2 20000000 > "$OUT/pmc_c2.log" 2>&1
grep -v "^copy" "$OUT/pmc_c2/summary.txt"
bash tools/run_pmc.sh "$OUT/pmc_c4" c4 8000000 > "$OUT/pmc_c4.log" 2>&1
grep -v "^copy" "$OUT/pmc_c4/summary.txt"
PMC_SQ_ONLY=1 bash tools/run_pmc.sh "$OUT/pmc_rag" rag 10000000 > "$OUT/pmc_rag.log" 2>&1
grep -v "^copy" "$OUT/pmc_rag/summary.txt"
# keep only small summaries in the merge-back
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
du -sh "$OUT"

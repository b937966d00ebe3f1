#!/bin/bash
# Round profile: bench line + ONE rocprofv3 --kernel-trace --stats process PER CONFIG (equal launches inside a config, so
# that the kernel's average duration in its csv is the duration of one launch of known size), then the PMC passes.
# usage (on the GPU box): bash tools/profile_round.sh r04
# Round 4: bench.py pins the launches per step of the configs whose stream does not fit the device (c3: 2, c4: 4), so the
# driver's run and these runs launch the same shapes; next to rocprofv3's own kernel_stats (every launch of the process)
# kernel_stats_<cfg>_full.csv holds the FULL-SIZE launches only (tools/kernel_stats_filter.py: no autotune trials, no spot
# checks, no cold first launch) -- every secondary.*.frac is avg_ms of that file, times launches_per_step.
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 1500 "$OUT/bench_default.json"; echo
# the headline config under rocprofv3: the driver's own command
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > "$OUT/bench_c2_under_rocprof.json" 2> "$OUT/trace_c2.err"
# the secondary configs, each alone, launches of one size (chunk-reads divides the reads)
for spec in "c3:0" "c4:0" "ref:0" "var:0" "var_slots:0" "c2_packed:0" "c2_dirty:0" "c2_dirty_slots:0"; do
  cfg=${spec%%:*}; chunk=${spec##*:}
  extra=""; if [ "$chunk" != "0" ]; then extra="--chunk-reads $chunk"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -o kt -- python bench.py --config $cfg --steps 5 --warmup 2 $extra --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > "$OUT/bench_${cfg}_under_rocprof.json" 2> "$OUT/trace_$cfg.err"
done
for cfg in c2 c3 c4 ref var var_slots c2_packed c2_dirty c2_dirty_slots; do
  for f in $(find "$OUT/trace_$cfg" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_$cfg.csv"; done
  echo "== $cfg"; python tools/kernel_stats_filter.py "$OUT/trace_$cfg" "$OUT/kernel_stats_${cfg}_full.csv" | head -4
done
# the consumers, one process: every kernel of every call (bench.py's `consumers` object under rocprofv3)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_consumers" -o kt -- python bench.py --consumers-reads 20000000 > "$OUT/bench_consumers_under_rocprof.json" 2> "$OUT/trace_consumers.err"
for f in $(find "$OUT/trace_consumers" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_consumers.csv"; done
python tools/show_consumers.py "$OUT/bench_consumers_under_rocprof.json"
# a consumer: per-read minimizers (w = 10) of 20 M clean reads, of the same reads with an N here and there, and given by offsets
for shape in clean dirty var; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mz_$shape" -o kt -- python tools/minimizer_bench.py 20000000 10 $shape > "$OUT/minimizer_${shape}_under_rocprof.txt" 2> "$OUT/trace_mz_$shape.err"
  for f in $(find "$OUT/trace_mz_$shape" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_minimizers_$shape.csv"; done
  cat "$OUT/minimizer_${shape}_under_rocprof.txt"; head -6 "$OUT/kernel_stats_minimizers_$shape.csv" 2>/dev/null
done
if [ -z "${SKIP_PMC:-}" ]; then
bash tools/run_pmc.sh "$OUT/pmc_c2" c2 20000000 > "$OUT/pmc_c2.log" 2>&1
grep -v "^copy" "$OUT/pmc_c2/summary.txt"
bash tools/run_pmc.sh "$OUT/pmc_c4" c4 8000000 > "$OUT/pmc_c4.log" 2>&1
grep -v "^copy" "$OUT/pmc_c4/summary.txt"
PMC_SQ_ONLY=1 bash tools/run_pmc.sh "$OUT/pmc_rag" rag 10000000 > "$OUT/pmc_rag.log" 2>&1
grep -v "^copy" "$OUT/pmc_rag/summary.txt"
fi
# keep only small summaries in the merge-back
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
du -sh "$OUT"

#!/bin/bash
# Everything the round's notes and DESIGN.md cite, made on one GPU box from the tree as it is:
#   bash tools/evidence_round.sh r06      (then: python tools/collect_profiles.py r04 copies the summaries into profiles/)
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 3000 bash tools/profile_round.sh "$TAG" > "$OUT/profile_round.log" 2>&1
# consumers, whole calls
{ for s in clean dirty var; do timeout 300 python tools/minimizer_bench.py 20000000 0 $s; done; } > "$OUT/minimizers.txt" 2>&1
{ timeout 300 python tools/bloom_bench.py 20000000 1; timeout 300 python tools/bloom_bench.py 20000000 3; timeout 300 python tools/count_bench.py 20000000 1; } > "$OUT/bloom_bench.txt" 2>&1
{ timeout 300 python tools/bloom_one.py 20000000 0 3; timeout 300 python tools/bloom_one.py 20000000 1 3; } > "$OUT/bloom_one.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bloom_one" -o kt -- python tools/bloom_one.py 20000000 0 3 > /dev/null 2>&1
for f in $(find "$OUT/trace_bloom_one" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_bloom_one.csv"; done
timeout 300 python tools/minhash_bench.py > "$OUT/minhash_bench.txt" 2>&1
# round 5: the binned read side of the filter / the sketch against the direct kernels, and its kernels under rocprofv3
timeout 600 python tools/query_bench.py 20000000 1,3 35 30 > "$OUT/query_bench.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_query" -o kt -- python tools/query_bench.py 20000000 1 35 0 > /dev/null 2>&1
for f in $(find "$OUT/trace_query" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_query_bench.csv"; done
# round 6: fresh-process default lines (one plain allocation per buffer)
{ for i in 1 2 3 4 5 6; do python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline --no-peak --no-plain-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['value']/1e9,1), 'G k-mers/s  frac', round(d['roofline']['frac'],4))"; done; } > "$OUT/default_line_spread.txt" 2>&1
# the record-form minimizer kernel under the counters (SQ groups)
MZ_W=10 PMC_SQ_ONLY=1 timeout 1200 bash tools/run_pmc.sh "$OUT/pmc_mzw" mz 20000000 > "$OUT/pmc_mzw.log" 2>&1
# spaced seeds: the shapes of the round-3 table, long seeds, seeds of few runs
timeout 600 python tools/seed_sweep.py > "$OUT/seed_sweep.txt" 2>&1
SWEEP_SHAPES="250,128,1,1;300,160,1,1;250,96,1,1;250,31,5,1;250,31,8,1;150,48,3,1;150,64,3,1;250,128,1,3" timeout 600 python tools/seed_sweep.py > "$OUT/seed_sweep_long.txt" 2>&1
timeout 600 python tools/seed_roll_sweep.py > "$OUT/seed_roll_sweep.txt" 2>&1
# round 6: the specialised seed kernel (hiprtc) under the counters -- LDS bank conflicts of the segment layout -- and the
# reference's benchmark shape / a long k of the k-mer path (what limits them)
timeout 600 bash tools/pmc_any.sh "$OUT/pmc_seed_rnd6" psj python tools/px_probe.py 250 31 rnd6 1 4000000 > "$OUT/pmc_seed_rnd6.txt" 2>&1
timeout 600 bash tools/pmc_any.sh "$OUT/pmc_seed_k128" psj python tools/px_probe.py 250 128 rnd1 1 4000000 > "$OUT/pmc_seed_k128.txt" 2>&1
NTHIP_SEED_JIT=0 timeout 600 bash tools/pmc_any.sh "$OUT/pmc_seed_static" seed_ps python tools/px_probe.py 250 128 3 1 4000000 > "$OUT/pmc_seed_static.txt" 2>&1
PMC_SQ_ONLY=1 timeout 900 bash tools/run_pmc.sh "$OUT/pmc_ref" shape:100,64,3 20000000 > "$OUT/pmc_ref.log" 2>&1
timeout 600 bash tools/pmc_any.sh "$OUT/pmc_seed_insert" bloom_ python tools/seed_insert_one.py > /dev/null 2>&1; python tools/pmc_levels_summary.py "$OUT/pmc_seed_insert" > "$OUT/pmc_seed_insert.txt" 2>&1
PMC_SQ_ONLY=1 timeout 900 bash tools/run_pmc.sh "$OUT/pmc_k200" shape:250,200,1 10000000 > "$OUT/pmc_k200.log" 2>&1
timeout 300 python tools/extend_bench.py > "$OUT/extend_bench.txt" 2>&1
timeout 600 python tools/facade_bench.py > "$OUT/facade_bench.txt" 2>&1
timeout 600 python tools/fastq_bench.py 10000000 256 1 kmers gz > "$OUT/fastq_gz_bench.txt" 2>&1
timeout 600 python tools/fasta_bench.py 3000 24 gz > "$OUT/fasta_gz_bench.txt" 2>&1
timeout 600 python tools/shape_sweep.py > "$OUT/shape_sweep.txt" 2>&1
SWEEP_SHAPES="150,65,1;150,100,1;250,200,1;1000,500,1;150,80,2;300,128,1;10000,200,1;400,255,1;100,64,3;100,64,1;150,64,1" timeout 600 python tools/shape_sweep.py > "$OUT/shape_sweep_long_k.txt" 2>&1
# round 5: hash STREAMS through the regions (stream / offsets / spaced-seed query, seed insert), and the reads kernel without its stores / its hashing
timeout 600 python tools/stream_query_bench.py > "$OUT/stream_query_bench.txt" 2>&1
timeout 900 python tools/stress_stream_query.py 60 1 > "$OUT/stress_stream_query.txt" 2>&1
{ timeout 300 python tools/seed_query_loop.py; timeout 300 python tools/var_alloc_spread.py; timeout 300 python tools/probe_cost.py; } > "$OUT/alloc_effects.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_seed_insert" -o kt -- python tools/seed_insert_one.py > "$OUT/seed_insert_one.txt" 2>&1
for f in $(find "$OUT/trace_seed_insert" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats_seed_insert.csv"; done
timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> "$OUT/pytest_gpu.txt" 2>&1
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*_agent_info.csv" -delete
du -sh "$OUT"

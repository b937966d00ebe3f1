#!/usr/bin/env python3
"""Copy the summaries of a GPU evidence round (tools/evidence_round.sh <tag> -> gpurun_out/<tag>/) into profiles/, tracked.

    python tools/collect_profiles.py r04
Small text / csv / json files only; traces and databases stay in the scratch directory.
"""
import glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {  # scratch name -> tracked name
    "bench_default.json": f"{tag}_bench_default.json",
    "minimizers.txt": f"{tag}_minimizers.txt",
    "bloom_bench.txt": f"{tag}_bloom_bench.txt",
    "minhash_bench.txt": f"{tag}_minhash_bench.txt",
    "bloom_one.txt": f"{tag}_bloom_one.txt",
    "query_bench.txt": f"{tag}_query_bench.txt",
    "default_line_spread.txt": f"{tag}_default_line_spread_evidence_box.txt",
    "pmc_seed_rnd6.txt": f"{tag}_pmc_seed_rnd6_summary.txt",
    "pmc_seed_k128.txt": f"{tag}_pmc_seed_k128_summary.txt",
    "pmc_seed_static.txt": f"{tag}_pmc_seed_static_summary.txt",
    "pmc_seed_insert.txt": f"{tag}_pmc_seed_insert_summary.txt",
    "pmc_ref/summary.txt": f"{tag}_pmc_ref_summary.txt",
    "pmc_k200/summary.txt": f"{tag}_pmc_k200_summary.txt",
    "seed_sweep.txt": f"{tag}_seed_sweep.txt",
    "seed_sweep_long.txt": f"{tag}_seed_sweep_long.txt",
    "seed_roll_sweep.txt": f"{tag}_seed_roll_sweep.txt",
    "extend_bench.txt": f"{tag}_extend_bench.txt",
    "facade_bench.txt": f"{tag}_facade_bench.txt",
    "shape_sweep.txt": f"{tag}_shape_sweep.txt",
    "shape_sweep_long_k.txt": f"{tag}_shape_sweep_long_k.txt",
    "stream_query_bench.txt": f"{tag}_stream_query_bench.txt",
    "stress_stream_query.txt": f"{tag}_stress_stream_query.txt",
    "alloc_effects.txt": f"{tag}_alloc_effects.txt",
    "kernel_stats_seed_insert.csv": f"{tag}_kernel_stats_seed_insert.csv",
    "reads_kernel_ablation.txt": f"{tag}_reads_kernel_ablation.txt",
    "pytest_gpu.txt": f"{tag}_pytest_gpu.txt",
    "fastq_gz_bench.txt": f"{tag}_fastq_gz_bench.txt",
    "fasta_gz_bench.txt": f"{tag}_fasta_gz_bench.txt",
    "profile_round.log": f"{tag}_profile_round.log",
    "pmc_c2/summary.txt": f"{tag}_pmc_c2_summary.txt",
    "pmc_c4/summary.txt": f"{tag}_pmc_c4_summary.txt",
    "pmc_rag/summary.txt": f"{tag}_pmc_reads_summary.txt",
    "pmc_mzw/summary.txt": f"{tag}_pmc_mzw_summary.txt",
    "bench_consumers_under_rocprof.json": f"{tag}_bench_consumers_under_rocprof.json",
}
for f in glob.glob(os.path.join(src, "bench_*_under_rocprof.json")):
    names[os.path.basename(f)] = f"{tag}_" + os.path.basename(f)
for f in glob.glob(os.path.join(src, "kernel_stats_*.csv")):
    names[os.path.basename(f)] = f"{tag}_" + os.path.basename(f)
for f in glob.glob(os.path.join(src, "minimizer_*_under_rocprof.txt")):
    names[os.path.basename(f)] = f"{tag}_" + os.path.basename(f)
n = 0
for a, b in sorted(names.items()):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) < (2 << 20):
        shutil.copyfile(p, os.path.join(dst, b))
        n += 1
    else:
        print("missing or too big:", a)
print(n, "files copied to profiles/")

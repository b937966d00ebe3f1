"""GPU tests of bench.py itself and of the full-size workloads (run with -m gpu on an MI355X).

* `python bench.py --gpus 2` must really run two ranks (it launches them itself); on a 1-GPU box the two ranks
  share the GPU over a gloo group (NTHASH_BENCH_SHARE_GPU=1) -- the control flow, the sharding, the max-over-ranks
  time and the all-rank verification are the ones an 8-GPU run uses.
* BASELINE config 5's per-GPU shard (125 M x 150 bp) and the chunked full-size configs 3 / 4 are checked through the
  on-device checksum of the whole hash stream against the checksums the REAL reference produced for the same reads
  (tests/golden/bench_checksums.json, tests/golden/gen_bench_checksums.py).
Nothing here reads /root/reference.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def run_bench(*argv, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_self_launch():
    """--gpus 2 without a launcher: two ranks, each its own shard, whole-job value, every rank verified"""
    res = run_bench("--gpus", "2", "--reads", "2000000", "--steps", "2", "--warmup", "1", "--dist-consumer-reads", "1000000",
                    env={"NTHASH_BENCH_SHARE_GPU": "1"})
    assert res["n_gpus"] == 2
    # the N-GPU consumer line: a filter per rank, OR-merged over the ring (here: two ranks, gloo), every rank's k-mers found
    dc = res["dist_consumer"]
    assert dc["ok"] is True and len(set(dc["set_bits"])) == 1 and dc["set_bits"][0] > 0, dc
    assert dc["kmers"] == 2 * 1_000_000 * 120 and dc["merge_ms"] > 0 and dc["merge_GBps_per_rank"] > 0
    assert res["scaling"] == "weak"
    assert res["verified_vs_oracle"] is True
    assert len(res["per_rank_kmers_per_s"]) == 2 and all(v > 0 for v in res["per_rank_kmers_per_s"])
    # whole-job value = both ranks' k-mers over the max-over-ranks time
    kmers = 2 * 2 * 2_000_000 * 120
    assert abs(res["value"] - kmers / (res["ms_per_step"] * 2 * 1e-3)) / res["value"] < 1e-6
    assert res["roofline"]["frac"] > 0 and res["roofline"]["kernel"] == "kmer_runs_kernel"


def test_bench_two_ranks_at_a_real_shards_size():
    """--gpus 2 with 60 M reads per rank: the two ranks' buffers -- 2 x (9 GB of reads + 57.6 GB of
    hashes) -- fill most of one GPU's HBM the way a 125 M-read shard fills a GPU of its own, so that the first real
    multi-GPU run cannot die on a size-dependent path (64-bit offsets, one launch per step, the verify pass, the consumer
    line); the ranks share this box's one GPU over gloo"""
    n = 60_000_000
    res = run_bench("--gpus", "2", "--reads", str(n), "--steps", "2", "--warmup", "1", "--dist-consumer-reads", "4000000",
                    env={"NTHASH_BENCH_SHARE_GPU": "1"}, timeout=1500)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["verified_vs_oracle"] is True   # (spot compare with the oracle on every rank; no committed checksum of 60 M-read shards)
    assert res["config"]["reads_per_gpu"] == n and res["config"]["launches_per_step"] == 1
    assert len(res["per_rank_kmers_per_s"]) == 2 and all(v > 1e10 for v in res["per_rank_kmers_per_s"])
    kmers = 2 * 2 * n * 120
    assert abs(res["value"] - kmers / (res["ms_per_step"] * 2 * 1e-3)) / res["value"] < 1e-6
    assert res["dist_consumer"]["ok"] is True


def test_bench_refuses_more_gpus_than_visible():
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--reads", "1000"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "n_gpus" not in p.stdout
    assert "refusing" in (p.stdout + p.stderr)


def test_bench_rejects_world_size_mismatch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", "1000"], cwd=ROOT,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "n_gpus" not in p.stdout


def _entry(name, first, n):
    for e in load_golden("bench_checksums.json"):
        if (e["workload"], e["first_read"], e["n_reads"]) == (name, first, n):
            return e
    raise KeyError((name, first, n))


def test_bench_rccl_process_group_at_world_size_one():
    """NTHASH_BENCH_FORCE_DIST=1: the branch an N-GPU run takes -- init_process_group(backend="nccl") = RCCL, barrier,
    all_reduce(MAX) of the step time, all_gather of the per-rank rates and verdicts -- runs on this box's one GPU"""
    res = run_bench("--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-peak",
                    "--no-secondary", "--dist-consumer-reads", "1000000", env={"NTHASH_BENCH_FORCE_DIST": "1"})
    assert res["n_gpus"] == 1
    assert res["dist_consumer"]["ok"] is True and res["dist_consumer"]["hardware_note"]
    assert res["dist"]["process_group"] == "nccl (RCCL)" and res["dist"]["forced_at_world_1"] is True
    assert res["verified_vs_oracle"] is True and len(res["per_rank_kmers_per_s"]) == 1
    assert abs(res["per_rank_kmers_per_s"][0] - res["value"]) / res["value"] < 0.2  # (value: max-reduced wall time)


@pytest.mark.parametrize("rank", list(range(8)))
def test_config5_shard_full_size_checksum(ctx, rank):
    """125 M x 150 bp -- one GPU's share of the 1 G-read job -- hashed in one call; checksum of all 15 G hashes
    against the reference's for reads [rank*125 M, (rank+1)*125 M)"""
    n, L, k = 125_000_000, 150, 31
    want = _entry("c2", rank * n, n)
    nwin = L - k + 1
    d_in = ctx.malloc(n * L)
    d_out = ctx.malloc(n * nwin * 8)
    try:
        ctx.synth_reads_ptr(d_in, rank * n, n, L, 42)
        tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin)
        assert tot == want["total"] == n * nwin
        s, x = ctx.checksum_ptr(d_out, tot)
        assert (format(s, "016x"), format(x, "016x")) == (want["sum"], want["xor"])
    finally:
        ctx.free(d_in)
        ctx.free(d_out)


@pytest.mark.parametrize("config", ["c3", "c4", "ref", "var", "c2_packed", "var_slots", "c2_dirty", "c2_dirty_slots"])
def test_full_size_chunked_configs_checksum(config):
    """configs 3 / 4 (outputs of 384 / 528 GB: produced chunk by chunk into a ring), the reference's own
    benchmark shape at full size, the variable-length batch (20 M reads of 100-150 bp as spans, reads with an N) and
    config 2 from 2-bit packed input, through bench.py's code path: whole stream == the reference's checksum"""
    res = run_bench("--config", config, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-peak")
    assert res["verify"]["ok"] is True, res["verify"]
    assert res["verify"]["spot_vs_oracle"] is True
    assert res["config"]["launches_per_step"] >= (2 if config in ("c3", "c4") else 1)
    assert "REDUCED" not in res["config"]["workload"]


def test_bench_default_line_has_all_parts():
    """the default run at a reduced size: every object the contract names is there"""
    res = run_bench("--reads", "4000000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "200000")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in res, key
    r = res["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["peak_measured"] and 2000 < r["peak_measured"] < 8000
    # (round 5: one plain allocation per buffer is the default -- no candidates measured, so no separate plain-allocation pass)
    assert "hipMalloc" in res["config"]["placement"]["allocator"] and all(b["candidates_measured"] == 1 for b in res["config"]["placement"]["buffers"])
    assert res["summary"]["c2"][0] > 0
    assert res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["cores"] == 1
    assert "REDUCED" in res["config"]["workload"]
    assert res["verified_vs_oracle"] is True


def test_bench_consumers_section():
    """bench.py's `consumers` object (Bloom insert / query, counting sketch, minimizers, MinHash on device-resident reads),
    at a tenth of its size: every check it carries holds"""
    res = run_bench("--consumers-reads", "2000000")["consumers"]
    for key in ("bloom_insert_fresh_4GiB", "count_insert_fresh_1Gi_counters", "minimizers_w10", "minimizers_w10_offsets", "minhash_m4",
                "bloom_query_4GiB", "count_query_1Gi_counters", "seed_bloom_insert_c4_seeds"):
        assert res[key]["ok"] is True and res[key]["value"] > 0, (key, res[key])
        r = res[key]["roofline"]    # (round 4: every consumer says what it must move and what its kernel of record took)
        assert r["algorithmic_bytes"] > 0 and 0 < r["frac"] < 1 and r["kernel"] and r["kernel_ms"] > 0, (key, r)
    q = res["bloom_query_4GiB"]     # (round 5: the binned query, the kernel it replaced beside it, what it moves through its lists)
    assert q["roofline"]["kernel"].startswith("bloom binned query") and q["roofline"]["list_traffic"]["GBps"] > 0
    assert q["direct_kernel"]["same_answer"] is True and q["direct_kernel"]["value"] > 0
    assert res["bloom_query_4GiB_m3"]["value"] > 0 and res["seed_bloom_insert_c4_seeds"]["query"]["value"] > 0
    assert res["minimizers_w10"]["roofline"]["kernel"] == "minimizer_w_kernel"

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
NTHASH_AMD_LIB=$PWD/nthash_amd/lib/ab/libnthash_hip_chunked.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 > gpurun_out/r02/bench_c2.json 2> gpurun_out/r02/bench_c2.err; python - <<'PY'
import json
r = json.load(open("gpurun_out/r02/bench_c2.json"))
print(r["value"]/1e9, r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("peak_measured"), r["verify"]["ok"], {k: (v.get("value",0)/1e9, v.get("frac"), v.get("verify_ok")) for k, v in r["secondary"].items()})
print(r["cpu_baseline"])
PY
timeout 300 python tools/ab_multi.py chunked,chunked:NTHIP_TUNE_PACING=1,:NTHIP_TUNE_WAVES=8 100000000 5 2>&1 | tail -5

#!/bin/bash
# the several-hashes copy-out without the division (multi_hash_copy_out): parity subset, then before / after on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/mh
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -n 4 2>&1 | tail -4
for cfg in ref; do
  for lib in before after mh2 mhplain before after mh2 mhplain; do
    if [ "$lib" = after ]; then unset NTHASH_AMD_LIB; else export NTHASH_AMD_LIB=$GRAFT_REPO_ROOT/nthash_amd/lib/ab/libnthash_hip_$lib.so; fi
    python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > gpurun_out/mh/${cfg}_$lib.json 2> gpurun_out/mh/${cfg}_$lib.err
    python - gpurun_out/mh/${cfg}_$lib.json "$cfg $lib" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:20s}: {d['value']/1e9:7.1f} G k-mers/s, kernel {r.get('kernel_avg_ms'):.3f} ms, frac {r['frac']:.4f}, verify {d.get('verify',{}).get('ok')}")
PY
  done
done

#!/bin/bash
# bench --config ref under a few run-length / wave settings (forward-half tables on)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/refk
for spec in "default" "NTHIP_TUNE_RUN_LEN=13" "NTHIP_TUNE_RUN_LEN=13 NTHIP_TUNE_WAVES=12" "NTHIP_TUNE_WAVES=8" "NTHIP_TUNE_RUN_LEN=10" "NTHIP_TUNE_WAVES=11" "NTHIP_TUNE_WAVES=10"; do
  tag=$(echo "$spec" | tr ' =' '__')
  if [ "$spec" = default ]; then envs=""; else envs="$spec"; fi
  env $envs python bench.py --config ref --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > gpurun_out/refk/$tag.json 2> gpurun_out/refk/$tag.err
  python - gpurun_out/refk/$tag.json "$spec" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:45s}: {d['value']/1e9:7.1f} G k-mers/s, kernel {r.get('kernel_avg_ms'):.3f} ms, frac {r['frac']:.4f}, verify {d.get('verify',{}).get('ok')}")
PY
done

#!/bin/bash
# reads per tile of kmer_reads_kernel on the variable-length bench batches (NTHIP_TUNE_READS_PER_TILE; default 32)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/varrpt
for cfg in var var_slots; do
  for r in 0 "$@"; do
    if [ "$r" = 0 ]; then unset NTHIP_TUNE_READS_PER_TILE; else export NTHIP_TUNE_READS_PER_TILE=$r; fi
    python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > gpurun_out/varrpt/${cfg}_$r.json 2>gpurun_out/varrpt/${cfg}_$r.err
    python - gpurun_out/varrpt/${cfg}_$r.json $cfg $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:10s} R={sys.argv[3]:>3s}: {d['value']/1e9:7.1f} G k-mers/s whole call, {d['ms_per_step']:.3f} ms/step, pass {r.get('kernel_avg_ms')} ms, verify {d.get('verify',{}).get('ok')}")
PY
  done
done

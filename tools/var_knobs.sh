#!/bin/bash
# waves per CU / run length of kmer_reads_kernel on the variable-length bench batches (NTHIP_TUNE_READS_WAVES, _READS_RUN_LEN)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/vark
for cfg in var_slots var; do
  for spec in "default" "NTHIP_TUNE_READS_WAVES=12" "NTHIP_TUNE_READS_WAVES=8" "NTHIP_TUNE_READS_RUN_LEN=11" "NTHIP_TUNE_READS_RUN_LEN=13" "NTHIP_TUNE_READS_RUN_LEN=7" "NTHIP_TUNE_READS_RUN_LEN=13 NTHIP_TUNE_READS_WAVES=12"; do
    tag=${cfg}_$(echo "$spec" | tr ' =' '__')
    if [ "$spec" = default ]; then envs=""; else envs="$spec"; fi
    env $envs python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-peak --no-plain-pass > gpurun_out/vark/$tag.json 2> gpurun_out/vark/$tag.err
    python - gpurun_out/vark/$tag.json "$cfg $spec" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:62s}: {d['value']/1e9:7.1f} G k-mers/s whole call, pass {r.get('kernel_avg_ms'):.3f} ms, verify {d.get('verify',{}).get('ok')}")
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
  done
done

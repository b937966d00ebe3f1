#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_var.json 2> gpurun_out/r02/bench_var.err
tail -c 400 gpurun_out/r02/bench_var.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_var.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
for k,v in d['secondary'].items(): print(k, {x:v.get(x) for x in ('value','frac','kernel','verify_ok','spot_vs_oracle','ms_per_step','error')})
PY

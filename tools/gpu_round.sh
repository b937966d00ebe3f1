#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['peak_measured'], d['verify']['ok'], [(b['fill_GBps'],b['candidates_measured']) for b in d['config']['placement']['buffers']])
for k,v in d['secondary'].items(): print(k, {x:v.get(x) for x in ('value','frac','verify_ok','ms_per_step','launches_per_step','placement_fill_GBps','error')})
PY
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -2

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-slots}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "read_slots or whole_read or fastx" 2>&1 | tail -12 | tee $OUT/pytest.log
for cfg in var var_slots; do
python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-peak --no-plain-pass > $OUT/$cfg.json 2> $OUT/$cfg.err
python - $OUT/$cfg.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]/1e9,1), "ms/step", round(d["ms_per_step"],3), "kernel", d["roofline"]["kernel"], round(d["roofline"]["kernel_avg_ms"],3), round(d["roofline"]["frac"],4), d["verify"]["ok"], d["verify"]["spot_vs_oracle"], d["verify"].get("total"), d["verify"].get("slots"))
PY
done

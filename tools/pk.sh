#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pk}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "packed or first_window or fixed or dirty or any_k" 2>&1 | tail -6 | tee $OUT/pytest.log
python bench.py --config c2_packed --steps 10 --warmup 3 --no-cpu-baseline --no-peak --no-plain-pass > $OUT/c2_packed.json 2> $OUT/c2_packed.err
python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/pk1/c2_packed.json").read().strip().splitlines()[-1])
print("c2_packed", round(d["value"]/1e9,1), round(d["roofline"]["frac"],4), d["roofline"]["kernel"], d["verify"]["ok"], d["verify"]["spot_vs_oracle"])
PY
tail -3 $OUT/c2_packed.err

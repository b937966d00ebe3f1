#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SWEEP_GIB=16 timeout 900 python tools/seed_sweep.py gpurun_out/seed_sweep.json 2>&1 | tee gpurun_out/seed_sweep.txt | tail -16
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_final.err | tee gpurun_out/bench_final.json | cut -c1-200

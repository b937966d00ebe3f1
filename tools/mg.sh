#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-mg}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "multi_device or fastx or packed" 2>&1 | tail -12 | tee $OUT/pytest.log

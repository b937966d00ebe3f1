set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
python tools/stress_seeds.py 2>&1 | tail -3

set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_facade.py -m gpu -x -q 2>&1 | tail -5
NTHASH_AMD_FORCE_DEVICE=0 ./oracle/_ref/ref_benchmark_on_facade

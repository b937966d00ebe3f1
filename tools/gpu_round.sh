cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
RAGGED_MOSTLY=150 python tools/ragged_bench.py 2>&1 | tail -2
RAGGED_MOSTLY=150 NTHIP_TUNE_NO_ROWS=1 python tools/ragged_bench.py 2>&1 | tail -2
RAGGED_MOSTLY=151 python tools/ragged_bench.py 2>&1 | tail -2
python tools/fastq_bench.py 2>&1 | tail -2
NTHIP_TUNE_NO_ROWS=1 python tools/fastq_bench.py 2>&1 | tail -2

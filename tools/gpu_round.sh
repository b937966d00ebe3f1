set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/ragged_bench.py 2>&1 | tail -4

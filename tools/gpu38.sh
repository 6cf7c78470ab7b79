cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_knapsack.py -x -q -m gpu --durations=12 2>&1 | tail -20

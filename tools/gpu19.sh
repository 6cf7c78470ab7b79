cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3

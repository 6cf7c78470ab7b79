cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_search" 2>&1 | tail -40

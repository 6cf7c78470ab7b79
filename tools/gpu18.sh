cd $GRAFT_REPO_ROOT
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_lazy" 2>&1 | grep -E "Error|error|assert" | head -8

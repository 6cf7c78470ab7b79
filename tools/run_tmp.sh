cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
DDO_HIP_ENGINE=1 timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(golden or replay or sequential_parity) and not dense and not tier" 2>&1 | tail -3

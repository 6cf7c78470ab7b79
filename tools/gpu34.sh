cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
DDO_HIP_LEX_CAP=3 timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or replay" 2>&1 | tail -3
bash tools/gpu26.sh

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
DDO_HIP_ENGINE=1 timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or replay or sequential_parity" 2>&1 | tail -3
python bench.py --no-cpu | cut -c1-330

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 1 2 3; do timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1; done
python bench.py | cut -c1-2400

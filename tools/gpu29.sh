cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu28.sh
python bench.py --no-cpu --concurrent 4096 | cut -c1-330
python bench.py --no-cpu --concurrent 1024 --steps 8 | cut -c1-330

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -s KILL 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | tail -2
DDO_HIP_ENGINE=1 timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | tail -2

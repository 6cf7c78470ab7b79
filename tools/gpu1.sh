set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -6
nproc; free -g | head -2
timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20
timeout -s KILL 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -40

cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -1 gpurun_out/profile_round.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
DDO_HIP_TIMES=1 timeout -s KILL 100 python -m pytest tests/test_gpu_api_surface.py -x -q -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundaries.py tests/test_gpu_distributed.py -x -q -p no:cacheprovider > gpurun_out/pytest_subset.log 2>&1; tail -2 gpurun_out/pytest_subset.log | cut -c1-200
echo finished

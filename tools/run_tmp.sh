cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 290 python -m pytest tests/test_gpu_knapsack.py tests/test_gpu_mcp.py tests/test_gpu_cache.py tests/test_gpu_max2sat.py tests/test_gpu_tsptw.py tests/test_gpu_vector_parity.py -x -q -p no:cacheprovider > gpurun_out/pytest_subset2.log 2>&1; tail -3 gpurun_out/pytest_subset2.log | cut -c1-200

cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
timeout -s KILL 1500 python -m pytest tests/test_gpu_tsptw.py tests/test_gpu_cache.py tests/test_gpu_knapsack.py -x -q -m gpu 2>&1 | tail -3
timeout -s KILL 900 python tools/tsptw_big.py 64 32 > gpurun_out/r04d/tsptw_beyond_64.jsonl 2> gpurun_out/r04d/tsptw_big.err; cat gpurun_out/r04d/tsptw_beyond_64.jsonl | cut -c1-600; tail -3 gpurun_out/r04d/tsptw_big.err
DDO_HIP_FIXED_LAYERS=1 timeout -s KILL 600 python tools/tsptw_big.py 64 32 --small 2>/dev/null | cut -c1-330

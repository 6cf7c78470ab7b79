# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "== new test"; timeout -s KILL 300 python -m pytest tests/test_gpu_tsptw.py -x -q -m gpu -p no:cacheprovider -k "global_memory" 2>&1 | tail -2
for t in 512 1024; do
echo "== DDO_HIP_THREADS=$t"; DDO_HIP_THREADS=$t timeout -s KILL 900 python -m pytest tests/test_gpu_mcp.py tests/test_gpu_max2sat.py tests/test_gpu_knapsack.py tests/test_gpu_vector_parity.py tests/test_gpu_tsptw.py -x -q -m gpu -p no:cacheprovider -k "not beyond_64 and not sized_like" 2>&1 | tail -2
done

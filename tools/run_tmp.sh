# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 200 python -m pytest tests/test_gpu_boundary_b1.py -x -q -m gpu -p no:cacheprovider -s -k "concurrent_compiles" 2>&1 | grep -o "T=[0-9].*\|[0-9]* passed.*\|[0-9]* failed.*" > gpurun_out/boundary_b1_threads.txt; cat gpurun_out/boundary_b1_threads.txt | cut -c1-250

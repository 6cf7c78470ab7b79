# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 500 python -m pytest tests/test_gpu_pooled.py -q -m gpu -p no:cacheprovider --tb=short -k "not golden and not replay" --timeout=100 --durations=6 > gpurun_out/pytest_pooled.log 2>&1; grep -v "^$" gpurun_out/pytest_pooled.log | tail -40 | cut -c1-400

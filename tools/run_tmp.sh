# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_gpu_api_surface.py tests/test_gpu_pooled.py -q -m gpu -p no:cacheprovider --tb=short -k "cutoffs_of_concurrent or weighted or raised_flag" --timeout=200 2>&1 | tail -15 | cut -c1-300

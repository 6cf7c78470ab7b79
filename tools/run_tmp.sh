# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=12 --timeout=300 > gpurun_out/pytest_full.log 2>&1; tail -22 gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -1 gpurun_out/profile_round.log

# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -2 gpurun_out/profile_round.log | cut -c1-200

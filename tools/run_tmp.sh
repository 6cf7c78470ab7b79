cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log

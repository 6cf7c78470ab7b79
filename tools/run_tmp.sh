# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== full gpu suite"; timeout -s KILL 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout 900 > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log
echo "== smoke"; timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for w in mcp max2sat tsptw; do
echo "== $w"; timeout -s KILL 300 python bench.py --workload $w 2>&1 | tail -1 > gpurun_out/sec_$w.json; cut -c1-200 gpurun_out/sec_$w.json
done
timeout -s KILL 300 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 2>&1 | tail -1 > gpurun_out/sec_frb15.json; cut -c1-200 gpurun_out/sec_frb15.json
timeout -s KILL 300 python bench.py --workload tsptw --instance AFG/rbg125a.tw --no-cpu 2>&1 | tail -1 > gpurun_out/sec_rbg125a.json; cut -c1-200 gpurun_out/sec_rbg125a.json
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log

cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -1 gpurun_out/profile_round.log
mkdir -p gpurun_out/r04f
for w in tsptw mcp max2sat; do timeout -s KILL 300 python bench.py --workload $w > gpurun_out/r04f/bench_$w.json 2>/dev/null; done
timeout -s KILL 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 --no-cpu > gpurun_out/r04f/bench_max2sat_frb15.json 2>/dev/null
timeout -s KILL 600 python tools/tsptw_big.py 64 32 > gpurun_out/r04f/tsptw_beyond_64.jsonl 2>/dev/null
echo finished

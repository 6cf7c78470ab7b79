cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for w in tsptw mcp max2sat; do timeout -s KILL 300 python bench.py --workload $w > gpurun_out/r04a/bench_$w.json 2> gpurun_out/r04a/bench_$w.err; tail -c 900 gpurun_out/r04a/bench_$w.json; echo; done
timeout -s KILL 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 --no-cpu > gpurun_out/r04a/bench_max2sat_frb15.json 2>/dev/null; tail -c 600 gpurun_out/r04a/bench_max2sat_frb15.json; echo
timeout -s KILL 1000 python tools/fringe_dups.py > gpurun_out/r04a/fringe_dups.jsonl 2> gpurun_out/r04a/fringe_dups.err; cat gpurun_out/r04a/fringe_dups.jsonl

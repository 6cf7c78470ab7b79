# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 900 python bench.py > gpurun_out/bench_stamped.json 2> gpurun_out/bench_stamped.err; tail -c 300 gpurun_out/bench_stamped.json

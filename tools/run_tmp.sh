# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 400 python tools/shim_bench.py brock400_1 10000 20 64 512 2048 > gpurun_out/shim_bench.jsonl 2> gpurun_out/shim_bench.err; cat gpurun_out/shim_bench.jsonl | cut -c1-400; tail -3 gpurun_out/shim_bench.err | cut -c1-200

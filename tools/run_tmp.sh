# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 400 python tools/micro/layer_bench.py --quick > gpurun_out/micro_layers.jsonl 2> gpurun_out/micro_layers.err; wc -l gpurun_out/micro_layers.jsonl; tail -2 gpurun_out/micro_layers.err

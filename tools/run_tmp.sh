# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
DDO_BENCH_ONE_GPU=1 timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 4 --warmup 2 --prove 300 > gpurun_out/bench_2ranks_one_gpu.json 2> gpurun_out/bench_2ranks_one_gpu.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/bench_2ranks_one_gpu.json") if l.startswith("{")][-1])
    print("2 ranks on one GPU: n_gpus", d["n_gpus"], "value %.3e" % d["value"], "scaling", d["scaling"], "proof", d.get("time_to_proved_optimum_s"), d["proof"]["proved"], d["proof"]["best_value"], "handed", d["proof"]["subproblems_handed_over"])
except Exception as e:
    print("2-rank bench FAILED", e)
PY
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=5 --timeout=300 > gpurun_out/pytest_full.log 2>&1; tail -9 gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -1 gpurun_out/profile_round.log | cut -c1-200

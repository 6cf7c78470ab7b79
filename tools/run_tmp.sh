# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
DDO_HIP_TIMES=1 timeout -s KILL 600 python bench.py --cpu-seconds 2 --b1-threads 0 > gpurun_out/bench_r05b.json 2> gpurun_out/bench_r05b.err; grep "ddo times" gpurun_out/bench_r05b.err | tail -7 | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r05b.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "ms", d["ms_per_step"])
print("proof", d.get("time_to_proved_optimum_s"), d["proof"]["roofline"]["frac"], [ (t["layer_capacity"], round(t["kernel_s"],1)) for t in d["proof"]["tiers_rank0"]])
PY

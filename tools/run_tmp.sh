# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
DDO_HIP_TIMES=1 timeout -s KILL 900 python bench.py --prove 0 --cpu-seconds 4 > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err; tail -5 gpurun_out/bench_b1.err | cut -c1-400
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_b1.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("boundary_b1"), indent=1)[:1800])
PY

cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
timeout -s KILL 900 python -m pytest tests/test_gpu_vector_parity.py tests/test_gpu_max2sat.py tests/test_gpu_mcp.py -x -q -m gpu 2>&1 | tail -2
for w in mcp max2sat; do timeout -s KILL 300 python bench.py --workload $w > gpurun_out/r04c/bench_$w.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r04c/bench_$w.json").read().strip().splitlines()[-1])
print("$w", "value %.4g" % d["value"], "frac %.5f" % d["roofline"]["frac"], "kernel_s %.4f wall_s %.4f" % (d["roofline"]["kernel_s"], d["roofline"]["wall_s"]), "speedup", d.get("speedup_vs_cpu"))
PY
done
timeout -s KILL 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 --no-cpu > gpurun_out/r04c/bench_max2sat_frb15.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r04c/bench_max2sat_frb15.json").read().strip().splitlines()[-1])
print("frb15 value %.4g frac %.4f" % (d["value"], d["roofline"]["frac"]))
PY
DDO_HIP_STATS=1 timeout 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 10 --no-cpu 2>&1 >/dev/null | grep "kcycles per layer" | tail -1 | cut -c1-330

#!/bin/bash
# round 2, GPU run 9: dense tier at W = 10000 (A/B on the frozen bench), proof time with it, parity suites
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run9; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tsptw.py tests/test_gpu_cache.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for cfg in "nodense:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --no-cpu > $O/ab_${name}.json 2> $O/ab_${name}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_run9/ab_*.json")):
    try:
        j=json.load(open(f)); r=j["roofline"]; print(f.split("/")[-1], "%.4g nodes/s"%j["value"], "ms/step %.2f"%j["ms_per_step"], "frac %.3f"%r["frac"], r["kernel"][:50], "kernel ms %.2f"%r["kernel_ms_avg"])
        for t in r.get("tiers", []): print("    ", t["kernel"][:60], "ms %.1f"%t["kernel_ms"], "launches", t["launches"], "subs", t["subproblems"], "up", t["handed_up"], "nodes %.3g"%t["nodes_expanded"])
    except Exception as e: print(f, "ERR", e)
PY
DDO_HIP_STATS=1 timeout 900 python bench.py --cpu-seconds 6 > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json; grep "tier [0-9]: layer" $O/bench.err | tail -8

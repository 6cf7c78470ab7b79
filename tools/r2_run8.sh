#!/bin/bash
# round 2, GPU run 8: valid dense A/B (launch now uses the kernel init picked), tier kernel actually used, arena growth
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run8; rm -rf $O; mkdir -p $O
for W in 2700; do
for cfg in "base:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env DDO_HIP_TIERS=0 $envs timeout 300 python bench.py --no-cpu --width $W --freeze-stride 2 > $O/ab_${name}_$W.json 2> $O/ab_${name}_$W.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_run8/ab_*.json")):
    try:
        j=json.load(open(f)); print(f.split("/")[-1], "%.4g nodes/s"%j["value"], "ms/step %.2f"%j["ms_per_step"], "kernel ms %.2f"%j["roofline"]["kernel_ms_avg"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests/test_gpu_tsptw.py tests/test_gpu_parity.py tests/test_gpu_cache.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
DDO_HIP_STATS=1 timeout 900 python bench.py --cpu-seconds 6 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; grep "tier" $O/bench.err | tail -8

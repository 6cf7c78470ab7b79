#!/bin/bash
# round 2, GPU run 7: dense (2 workgroups per CU) A/B at a width whose table fits twice, then the full gpu suite and the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run7; rm -rf $O; mkdir -p $O
for W in 2700 2000; do
for cfg in "base:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1" "t512:DDO_HIP_THREADS=512"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env DDO_HIP_TIERS=0 $envs timeout 300 python bench.py --no-cpu --width $W --freeze-stride 2 > $O/ab_${name}_$W.json 2> $O/ab_${name}_$W.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_run7/ab_*.json")):
    try:
        j=json.load(open(f)); print(f.split("/")[-1], "%.4g nodes/s"%j["value"], "ms/step %.2f"%j["ms_per_step"], "kernel ms %.2f"%j["roofline"]["kernel_ms_avg"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json

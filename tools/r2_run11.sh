#!/bin/bash
# round 2, GPU run 12: shortened expand chain (free slot + path with the first loads, paired inserts): parity, probes, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run11; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for cfg in "nodense:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --no-cpu > $O/ab_${name}.json 2> $O/ab_${name}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_run11/ab_*.json")):
    try:
        j=json.load(open(f)); r=j["roofline"]; print(f.split("/")[-1], "%.4g nodes/s"%j["value"], "ms/step %.2f"%j["ms_per_step"], "frac %.3f"%r["frac"], r["kernel"][:50], "kernel ms %.2f"%r["kernel_ms_avg"])
    except Exception as e: print(f, "ERR", e)
PY
mv ddo_amd/_build ddo_amd/_build_base; cp -r ddo_amd/_build_probes ddo_amd/_build
for cfg in "nodense:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs DDO_HIP_STATS=1 timeout 300 python bench.py --no-cpu > $O/probe_${name}.json 2> $O/probe_${name}.err
  grep -h "kcycles per layer\|per layer: lex" $O/probe_${name}.err | tail -2 | cut -c1-1200
done
rm -rf ddo_amd/_build; cp -r ddo_amd/_build_deep ddo_amd/_build
DDO_HIP_DENSE=0 timeout 300 python bench.py --no-cpu > $O/ab_deep1024.json 2> $O/ab_deep1024.err
python -c "
import json; j=json.load(open('$O/ab_deep1024.json')); print('deep1024 %.4g nodes/s ms/step %.2f'%(j['value'], j['ms_per_step']))"
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

#!/bin/bash
# round 2, GPU run 10: dense-tier parity, expand-chain probes (probes build, full engine vs dense), proof time with the dense tier
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dense or tiers" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
mv ddo_amd/_build ddo_amd/_build_base; cp -r ddo_amd/_build_probes ddo_amd/_build
for cfg in "nodense:DDO_HIP_DENSE=0" "dense:DDO_HIP_DENSE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs DDO_HIP_STATS=1 timeout 300 python bench.py --no-cpu > $O/probe_${name}.json 2> $O/probe_${name}.err
  grep -h "kcycles per layer\|per layer: lex" $O/probe_${name}.err | tail -2 | cut -c1-1200
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build
timeout 900 python bench.py --cpu-seconds 6 > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json

#!/bin/bash
# Runs on the GPU box: quick parity subset of the in-place kernels, then the frozen bench three times.
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or shrunk or maximum or lazy or tiers or dense or full_size or (replay_of_oracle and (brock200_2 or keller4))" 2>&1 | tail -4
for rep in 1 2 3; do
  timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms/step %.2f frac %.4f kernel_ms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg']))"
done
if [ -n "$STATS" ]; then DDO_HIP_STATS=1 python bench.py --no-cpu 2>&1 >/dev/null | grep "ddo stats" | grep -v "launch of" | head -12; fi
if [ -n "$PROBES" ] && [ -d ddo_amd/_build_probes ]; then
  mv ddo_amd/_build ddo_amd/_build_base; cp -r ddo_amd/_build_probes ddo_amd/_build
  DDO_HIP_STATS=1 python bench.py --no-cpu 2>&1 >/dev/null | grep "ddo stats" | grep "per layer:" | sed "s/.*expand chain/expand chain/"
  rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build
fi

#!/bin/bash
# round 2, GPU run 14: same-box A/B of two library builds (a66f16b expand chain vs the reworked one), dense off and on
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run14; rm -rf $O; mkdir -p $O
mv ddo_amd/_build ddo_amd/_build_new
for rep in 1 2; do
 for v in _build_old _build_new; do
  for d in 0 1; do
   rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
   echo "$v dense=$d: $(DDO_HIP_DENSE=$d timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1)" | tee -a $O/ab.txt
  done
 done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_new ddo_amd/_build

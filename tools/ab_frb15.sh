#!/bin/bash
# same-box A/B of two library builds on the layer-rebuilding engine's largest workload (MAX2SAT frb15-9-1, 30 s budget):
#   gpurun -- 'bash tools/ab_frb15.sh _build_old'     (directories under ddo_amd/; the current _build is the base)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/ab_frb15; mkdir -p $O; : > $O/ab.txt
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2; do
 for v in _build_base "$@"; do
  rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
  echo "$v: $(timeout -s KILL 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 --no-cpu 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1)" | tee -a $O/ab.txt
 done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

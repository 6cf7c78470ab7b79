#!/bin/bash
# same-box A/B of library builds: tools/ab_dirs.sh <dir> ... (directories under ddo_amd/), dense tier off and on, two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/ab_dirs; mkdir -p $O; : > $O/ab.txt
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2; do
 for v in _build_base "$@"; do
  for d in 0 1; do
   rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
   echo "$v dense=$d: $(DDO_HIP_DENSE=$d timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1)" | tee -a $O/ab.txt
  done
 done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

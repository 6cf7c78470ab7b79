#!/bin/bash
# same-box A/B of library builds on config C3 (MAX2SAT frb10-6-1, W = 5000): tools/ab_max2sat.sh <dir> ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/ab_max2sat; mkdir -p $O; : > $O/ab.txt
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2; do
 for v in _build_base "$@"; do
   rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
   echo "$v: $(timeout -s KILL 300 python bench.py --workload max2sat --no-cpu 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g nodes/s, proof %.3f s, kernel %.1f ms x %d, best %s'%(j['value'], j['time_to_proved_optimum_s'], j['roofline']['kernel_ms_avg'], j['roofline']['launches'], j['best_value']))")" | tee -a $O/ab.txt
 done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

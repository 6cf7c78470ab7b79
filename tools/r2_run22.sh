#!/bin/bash
# round 2, GPU run 22: tier hints per (depth, ub - incumbent) cell vs per depth -- time to proof A/B with per-tier statistics
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run22; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for hs in 0 1; do
  echo "hint_by_slack=$hs: $(DDO_HIP_HINT_SLACK=$hs DDO_HIP_STATS=1 timeout 400 python tools/search_stats.py brock400_1 10000 8192 300 2> $O/err_${hs}_$rep.txt | sed 's/{[^}]*}//')" | tee -a $O/proof_ab.txt
  grep "tier [0-9]: layer" $O/err_${hs}_$rep.txt | sed 's/LDS [0-9]* B | //' | tee -a $O/proof_ab.txt
done
done

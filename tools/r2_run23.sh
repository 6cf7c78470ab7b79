#!/bin/bash
# round 2, GPU run 23: at which retry share should a (depth, slack) cell skip a tier?  time to proof per threshold
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run23; rm -rf $O; mkdir -p $O
for pct in 50 70 90 30; do
  echo "tier_skip=$pct: $(DDO_HIP_TIER_SKIP=$pct DDO_HIP_STATS=1 timeout 400 python tools/search_stats.py brock400_1 10000 8192 300 2> $O/err_$pct.txt | sed 's/{[^}]*}//')" | tee -a $O/proof_ab.txt
  grep "tier [0-9]: layer" $O/err_$pct.txt | sed 's/LDS [0-9]* B | //' | cut -c1-200 | tee -a $O/proof_ab.txt
done

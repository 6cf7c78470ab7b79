#!/bin/bash
# round 2, GPU run 20: result records written straight to pinned host memory vs copied per launch -- time to proof A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run20; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for cp in 1 0; do
  echo "results_copy=$cp: $(DDO_HIP_RESULTS_COPY=$cp timeout 400 python tools/search_stats.py brock400_1 10000 8192 300 2>/dev/null | sed 's/{[^}]*}//')" | tee -a $O/proof_ab.txt
done
done

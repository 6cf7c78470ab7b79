#!/bin/bash
# round 2, GPU run 18: how does the frozen-bench rate scale with the number of decision diagrams in flight (full-width engine, 1 per CU)?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run18; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for s in 64 128 192 256; do
  echo "slots=$s: $(DDO_HIP_DENSE=0 DDO_HIP_TIERS=0 DDO_HIP_SLOTS=$s timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1)" | tee -a $O/slots.txt
done
for s in 256 384 512; do
  echo "dense slots=$s: $(DDO_HIP_DENSE=1 DDO_HIP_TIERS=0 DDO_HIP_TIER_SLOTS=$s timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | grep -o '"value": [0-9.e+]*' | head -1)" | tee -a $O/slots.txt
done
done

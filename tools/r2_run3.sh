#!/bin/bash
# round 2, GPU run 3: full GPU suite (tier fix, vector parity, distributed), tier configurations vs time to proof
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run3; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for cfg in "256:64,1024:128" "128:64,1024:128" "256:64,2048:256" "256:64,640:128"; do
  DDO_HIP_STATS=1 DDO_HIP_TIERS=$cfg timeout 600 python tools/search_stats.py brock400_1 10000 8192 600 > $O/proof_$cfg.log 2> $O/proof_$cfg.err
  echo "tiers $cfg: $(tail -1 $O/proof_$cfg.log | cut -c1-60) $(tail -1 $O/proof_$cfg.log | grep -o '} [0-9.]* device.*')"
  grep "ddo stats\] tier [0-9]: layer" $O/proof_$cfg.err
done
timeout 900 python tools/search_stats.py brock400_1 10000 16384 600 > $O/proof_16k.log 2> $O/proof_16k.err; tail -1 $O/proof_16k.log | grep -o '} [0-9.]* device.*'

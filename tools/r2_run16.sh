#!/bin/bash
# round 2, GPU run 16: validation of the round's final build -- full gpu suite, smoke, profile round (default bench + rocprofv3 passes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run16; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
timeout 600 python bench.py --workload max2sat > $O/bench_max2sat.json 2> $O/bench_max2sat.err

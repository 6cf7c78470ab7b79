#!/bin/bash
# round 2, GPU run 28: validation of the final tree -- full gpu suite, smoke, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run28; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.load(open('$O/bench.json')); print({k:j[k] for k in ('value','ms_per_step','speedup_vs_cpu','time_to_proved_optimum_s')}, j['roofline']['frac'], j['roofline']['kernel_ms_avg'])"

#!/bin/bash
# round 2, GPU run 17: layer-rebuilding engine after the tie-break change -- suites of every model on it, C3 bench, slowest tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run17; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --workload max2sat > $O/bench_max2sat.json 2> $O/bench_max2sat.err; python -c "
import json; j=json.load(open('$O/bench_max2sat.json')); print('max2sat %.4g nodes/s proof %.3f s frac %.4f kernel ms %.1f x%d'%(j['value'], j['time_to_proved_optimum_s'], j['roofline']['frac'], j['roofline']['kernel_ms_avg'], j['roofline']['launches']))"
timeout 2400 python -m pytest tests -m gpu -q --durations=40 --deselect tests/test_gpu_parity.py > $O/pytest.log 2>&1; tail -50 $O/pytest.log

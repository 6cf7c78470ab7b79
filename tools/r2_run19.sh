#!/bin/bash
# round 2, GPU run 19: phase clocks of the layer-rebuilding engine on config C3 (MAX2SAT frb10-6-1, W = 5000)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run19; rm -rf $O; mkdir -p $O
DDO_HIP_STATS=1 timeout 600 python bench.py --workload max2sat --no-cpu > $O/bench_max2sat.json 2> $O/bench_max2sat.err
grep -h "kcycles per layer\|DDs " $O/bench_max2sat.err | tail -4 | cut -c1-700
python -c "
import json; j=json.load(open('$O/bench_max2sat.json')); print('max2sat %.4g nodes/s proof %.3f s frac %.4f kernel ms %.1f x%d'%(j['value'], j['time_to_proved_optimum_s'], j['roofline']['frac'], j['roofline']['kernel_ms_avg'], j['roofline']['launches']))"
timeout 900 python -m pytest tests/test_gpu_cache.py tests/test_gpu_max2sat.py tests/test_gpu_knapsack.py -m gpu -q --durations=3 2>&1 | tail -8

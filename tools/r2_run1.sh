#!/bin/bash
# round 2, GPU run 1: new tests + frozen bench reproducibility + search statistics by DD size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_api_surface.py tests/test_gpu_parity.py -m gpu -x -q -k "concurrent or sharded or lazy or time_budget or set_primal" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
timeout 600 python bench.py --no-cpu --steps 7 --warmup 1 > $O/bench_7_1.json 2> $O/bench_7_1.err
python - <<'PY'
import json
for f in ("bench_default","bench_20_5","bench_7_1"):
    try:
        j=json.load(open(f"gpurun_out/r2_run1/{f}.json"))
        print(f, "%.4g nodes/s"%j["value"], "ms/step %.1f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
DDO_HIP_STATS=1 timeout 900 python tools/search_stats.py brock400_1 10000 8192 600 > $O/stats.log 2> $O/stats.err; tail -3 $O/stats.log; grep "ddo stats\] \(widest\|DDs\|slot\|device\)" $O/stats.err | tail -30

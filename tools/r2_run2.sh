#!/bin/bash
# round 2, GPU run 2: full GPU suite, frozen bench, tiers: search statistics + time to proof
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
timeout 600 python bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_3_1.json 2> $O/bench_3_1.err
python - <<'PY'
import json
for f in ("bench_default","bench_20_5","bench_3_1"):
    try:
        j=json.load(open(f"gpurun_out/r2_run2/{f}.json"))
        print(f, "%.4g nodes/s"%j["value"], "ms/step %.1f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"], "nodes/launch %.3g"%j["roofline"]["nodes_per_launch"], j["config"]["workload"][-120:])
    except Exception as e: print(f, "ERR", e)
PY
timeout 900 python tools/search_stats.py brock400_1 10000 8192 600 > $O/proof_tiers.log 2> $O/proof_tiers.err; tail -2 $O/proof_tiers.log
DDO_HIP_TIERS=0 timeout 900 python tools/search_stats.py brock400_1 10000 8192 600 > $O/proof_notiers.log 2> $O/proof_notiers.err; tail -2 $O/proof_notiers.log
DDO_HIP_STATS=1 timeout 900 python tools/search_stats.py brock400_1 10000 8192 600 > $O/stats.log 2> $O/stats.err; tail -2 $O/stats.log; grep "ddo stats\] \(tier [0-9]: layer\|widest\|DDs\|slot\|host\)" $O/stats.err | tail -30

#!/bin/bash
# round 2, GPU run 4: frontier/cache tests + full suite, 512-thread DEEP variant A/B, profile round for profiles/r02
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run4; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_cache.py -m gpu -x -q > $O/pytest_cache.log 2>&1; tail -6 $O/pytest_cache.log
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_cache.py > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for t in 1024 512; do
  DDO_HIP_THREADS=$t timeout 300 python bench.py --no-cpu > $O/bench_t$t.json 2> $O/bench_t$t.err
done
python - <<'PY'
import json
for f in ("bench_t1024","bench_t512"):
    try:
        j=json.load(open(f"gpurun_out/r2_run4/{f}.json"))
        print(f, "%.4g nodes/s"%j["value"], "ms/step %.1f"%j["ms_per_step"], "frac %.3f"%j["roofline"]["frac"], "kernel ms %.1f"%j["roofline"]["kernel_ms_avg"])
    except Exception as e: print(f, "ERR", e)
PY
bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log

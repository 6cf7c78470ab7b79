#!/bin/bash
# round 2, GPU run 6: TSPTW + dominance + cache suites, C3 bench line, micro-benchmark (quick grid)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run6; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tsptw.py tests/test_gpu_cache.py tests/test_cli.py -m gpu -q -x > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 600 python bench.py --workload max2sat > $O/bench_max2sat.json 2> $O/bench_max2sat.err; tail -c 900 $O/bench_max2sat.json
timeout 900 python tools/micro/layer_bench.py --quick > $O/micro_quick.jsonl 2> $O/micro_quick.err; tail -4 $O/micro_quick.jsonl

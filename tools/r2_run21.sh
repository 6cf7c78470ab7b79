#!/bin/bash
# round 2, GPU run 21: final validation -- full gpu suite, smoke, 2-rank bench on one GPU (the driver's N>1 launch line), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run21; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
DDO_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; tail -c 900 $O/bench_2ranks.json; tail -3 $O/bench_2ranks.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 500 $O/bench.json

cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06k; mkdir -p $O
timeout -s KILL 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
timeout 300 python bench.py --workload max2sat > $O/bench_max2sat.json 2>/dev/null
timeout 300 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 > $O/bench_max2sat_frb15-9-1.json 2>/dev/null
timeout 300 python bench.py --workload mcp > $O/bench_mcp.json 2>/dev/null
timeout 300 python bench.py --workload tsptw > $O/bench_tsptw.json 2>/dev/null
timeout 300 python bench.py --workload tsptw --instance AFG/rbg125a.tw > $O/bench_tsptw_rbg125a.json 2>/dev/null
timeout 400 python bench.py --workload misp-pooled > $O/bench_misp_pooled.json 2> $O/bench_misp_pooled.err
timeout 300 python tools/shim_bench.py brock400_1 10000 20 64 512 2048 > $O/shim_bench.jsonl 2> $O/shim_bench.err
timeout 300 python -m pytest tests/test_gpu_boundary_b1.py -q -m gpu -s 2>&1 | grep -E "T=[0-9]+:|passed|failed" > $O/boundary_b1_threads.txt
timeout 900 python tools/micro/layer_bench.py --quick > $O/micro_layers.jsonl 2> $O/micro.err
timeout 1500 bash tools/dist_overhead.sh > $O/dist_overhead.log 2>&1; cp gpurun_out/dist_overhead.jsonl $O/dist_overhead.jsonl
DDO_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
ls -la $O

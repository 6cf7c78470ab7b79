#!/bin/bash
# Search overhead of the hash-sharded best-first search, measured on ONE GPU (DDO_BENCH_ONE_GPU=1: every rank on cuda:0, gloo):
# sub-problems explored / nodes expanded / hand-overs with 1, 2, 4 and 8 ranks on the same instance, with rebalancing by the best
# open bound (--ub-gap 2, default) and without (--ub-gap 0).  The wall time is NOT a scaling figure (the ranks share one GPU).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/dist_overhead.jsonl
: > $OUT
run() {  # nproc, extra args...
  local n=$1; shift
  local port=$((20000 + RANDOM % 20000))
  DDO_BENCH_ONE_GPU=1 DDO_HIP_POOL_GB=8 timeout -s KILL 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $port \
      -m ddo_amd.dist_main "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); d['args'] = sys.argv[1]; print(json.dumps(d))" "$*" >> $OUT
}
for inst in "data/misp/brock200_4.clq -w 200 -t 256" "data/misp/p_hat300-1.clq -w 500 -t 256" "data/misp/brock200_1.clq -w 500 -t 512"; do
  for n in 1 2 4 8; do
    for gap in 2 0; do
      [ $n = 1 ] && [ $gap = 0 ] && continue
      run $n $inst --ub-gap $gap
    done
  done
done
cat $OUT

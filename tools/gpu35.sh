cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DDO_BENCH_ONE_GPU=1 DDO_HIP_POOL_GB=16 timeout -s KILL 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -5 | cut -c1-900
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -3

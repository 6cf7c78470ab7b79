set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 6 --warmup 2 --no-cpu 2>&1 | grep -E "ddo stats|value" | cut -c1-900
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 3 --warmup 2 --no-cpu --concurrent 2048 2>&1 | grep -E "ddo stats|value" | cut -c1-400
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 2 --no-cpu --concurrent 8192 2>&1 | grep -E "ddo stats|value" | cut -c1-400

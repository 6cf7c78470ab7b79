cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | grep -E "kcycles|per layer|\"value\"" | cut -c1-400

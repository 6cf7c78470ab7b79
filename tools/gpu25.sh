cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in 32 64 128 256; do
echo "== slots $S"
DDO_HIP_SLOTS=$S DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 2 --warmup 2 --no-cpu --concurrent 1024 2>&1 | grep -E "kcycles|\"value\"" | cut -c1-330
done

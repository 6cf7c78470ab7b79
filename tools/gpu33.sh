cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for T in 1024 512; do
echo "== threads $T"
DDO_HIP_THREADS=$T DDO_HIP_STATS=1 timeout -s KILL 300 python bench.py --no-cpu 2>&1 | grep -E "kcycles per layer: misc|\"value\"" | cut -c1-330
done

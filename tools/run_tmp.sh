cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() { echo "$1: $(env $2 timeout -s KILL 300 python bench.py --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|rror[^"]*' | head -3 | tr '\n' ' ')"; }
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2; do
rm -rf ddo_amd/_build; cp -r ddo_amd/_build_base ddo_amd/_build; run "base 2x512" "X=1"
rm -rf ddo_amd/_build; cp -r ddo_amd/_build_w8 ddo_amd/_build; run "w8 2x1024 (64 VGPRs)" "DDO_HIP_DENSE_THREADS=1024"
rm -rf ddo_amd/_build; cp -r ddo_amd/_build_w6 ddo_amd/_build; run "w6 2x768 (80 VGPRs)" "DDO_HIP_DENSE_THREADS=768"
done
rm -rf ddo_amd/_build; cp -r ddo_amd/_build_w8 ddo_amd/_build
DDO_HIP_STATS=1 DDO_HIP_DENSE_THREADS=1024 python bench.py --no-cpu 2>&1 >/dev/null | grep "ddo stats" | grep -E "tier 2:|kcycles per layer" | tail -3
timeout -s KILL 300 env DDO_HIP_DENSE_THREADS=1024 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and brock400 and dense" 2>&1 | tail -2
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

cd "$GRAFT_REPO_ROOT"
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or shrunk or maximum or lazy or tiers or dense or full_size or (replay_of_oracle and (brock200_2 or keller4 or brock400))" 2>&1 | tail -4
echo "no-stash: $(DDO_HIP_NO_WLKH=1 timeout -s KILL 300 python bench.py --no-cpu 2>&1 | grep -o '"value": [0-9.]*')"
bash tools/ab_builds.sh _build_g0
echo "no-stash: $(DDO_HIP_NO_WLKH=1 timeout -s KILL 300 python bench.py --no-cpu 2>&1 | grep -o '"value": [0-9.]*')"
DDO_HIP_STATS=1 python bench.py --no-cpu 2>&1 >/dev/null | grep "ddo stats" | grep -E "kcycles per layer|per layer:" 

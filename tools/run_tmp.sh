cd "$GRAFT_REPO_ROOT"
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or shrunk or dense or (replay_of_oracle and (brock200_2 or keller4))" 2>&1 | tail -4
bash tools/ab_builds.sh _build_g0 _build_g1 _build_g4

cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 python tools/run_tmp.py AFG/rbg132.tw AFG/rbg125a.tw 2>&1 | tail -10

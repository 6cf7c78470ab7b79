cd "$GRAFT_REPO_ROOT"
bash tools/ab_builds.sh _build_now

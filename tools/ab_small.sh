#!/bin/bash
# like tools/ab_builds.sh, for the small-DD regime: whole search of brock200_1 (W = 10 000, 4096 in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2 3; do
  for v in _build_base "$@"; do
    rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
    echo "$v: $(timeout -s KILL 300 python tools/search_stats.py brock200_1 10000 4096 2>&1 | grep -o "} [0-9.]* device ([0-9.]*")"
  done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

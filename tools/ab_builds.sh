#!/bin/bash
# A/B of library builds on ONE GPU box: tools/ab_builds.sh <dir> <dir> ...  (directories under ddo_amd/, e.g. _build _build_x);
# each is swapped in as ddo_amd/_build for three alternating bench runs.
cd "$GRAFT_REPO_ROOT" || exit 1
mv ddo_amd/_build ddo_amd/_build_base
for rep in 1 2 3; do
  for v in _build_base "$@"; do
    rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
    echo "$v: $(timeout -s KILL 300 python bench.py --no-cpu 2>&1 | grep -o '"value": [0-9.]*')"
  done
done

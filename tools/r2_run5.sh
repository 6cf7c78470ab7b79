#!/bin/bash
# round 2, GPU run 5: cache tests after the frontier-order fix, frb15 (72-word states), whole suite smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run5; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cache.py tests/test_gpu_vector_parity.py -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log

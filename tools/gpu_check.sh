#!/bin/bash
# One gpurun call that answers "is the tree still good?": full GPU parity suite (both models), the tie-break radix path
# and engine 1 forced, then the bench with per-phase statistics.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_check.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
DDO_HIP_LEX_CAP=3 timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or replay" 2>&1 | tail -2
DDO_HIP_ENGINE=1 timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(golden or replay or sequential_parity) and not dense and not tier" 2>&1 | tail -2
for cfg in DDO_HIP_THREADS=512 DDO_HIP_KEYS_GLOBAL=1 DDO_HIP_SLOTS=7; do   # alternate launch shapes: keys in L2, half-size workgroups, few slots
    env $cfg timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or replay or lazy" 2>&1 | tail -1
done
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --no-cpu 2>&1 | grep -E "kcycles|per layer|host s|\"value\"" | cut -c1-420

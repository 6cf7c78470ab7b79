cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu --concurrent 4096 2>&1 | grep -E "ddo stats|value|Error|error" | cut -c1-700
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | grep -E "value|Error|error" | cut -c1-700
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 8 --warmup 2 --no-cpu --concurrent 1024 2>&1 | grep -E "value|Error|error" | cut -c1-700

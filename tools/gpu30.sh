cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k golden > /tmp/t.log 2>&1
head -c 1500 /tmp/t.log
echo; echo ----; dmesg 2>/dev/null | tail -5

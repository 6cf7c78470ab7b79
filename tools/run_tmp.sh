# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
DDO_HIP_TIMES=1 timeout -s KILL 900 python -m pytest tests/test_gpu_boundary_b1.py -q -s -m gpu -p no:cacheprovider > gpurun_out/pytest_b1.log 2>&1; grep "T=\|ddo times\|passed\|failed\|Error" gpurun_out/pytest_b1.log | cut -c1-400

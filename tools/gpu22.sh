cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or lazy or replay" 2>&1 | tail -2
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu --concurrent 4096 2>&1 | grep -E "ddo stats. device kcyc|value|Error|error" | cut -c1-600
mkdir -p gpurun_out/prof4
B="python bench.py --steps 2 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --kernel-trace -d gpurun_out/prof4/tcp -o r --output-format csv -- $B > gpurun_out/prof4/tcp.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof4/write -o r --output-format csv -- $B > gpurun_out/prof4/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof4/fetch -o r --output-format csv -- $B > gpurun_out/prof4/fetch.log 2>&1

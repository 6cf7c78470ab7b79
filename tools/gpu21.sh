cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof3
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof3/fetch -o r --output-format csv -- $B > gpurun_out/prof3/fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof3/write -o r --output-format csv -- $B > gpurun_out/prof3/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --kernel-trace -d gpurun_out/prof3/tcc -o r --output-format csv -- $B > gpurun_out/prof3/tcc.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --kernel-trace -d gpurun_out/prof3/tcp -o r --output-format csv -- $B > gpurun_out/prof3/tcp.log 2>&1
grep -h value gpurun_out/prof3/fetch.log | cut -c1-200

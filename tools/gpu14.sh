cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 2 --no-cpu"
rocprofv3 -L 2>/dev/null | grep -E "^\s*(TCC_|TCP_|SQ_)" | awk '{print $1}' | sort -u | tr '\n' ' ' | head -c 6000 > gpurun_out/prof2/counters.txt
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof2/fetch -o r --output-format csv -- $B > gpurun_out/prof2/fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof2/write -o r --output-format csv -- $B > gpurun_out/prof2/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/prof2/sq -o r --output-format csv -- $B > gpurun_out/prof2/sq.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/prof2/sq2 -o r --output-format csv -- $B > gpurun_out/prof2/sq2.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --kernel-trace -d gpurun_out/prof2/tcc -o r --output-format csv -- $B > gpurun_out/prof2/tcc.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --kernel-trace -d gpurun_out/prof2/tcp -o r --output-format csv -- $B > gpurun_out/prof2/tcp.log 2>&1
tail -3 gpurun_out/prof2/tcc.log gpurun_out/prof2/tcp.log | cut -c1-300

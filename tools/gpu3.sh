set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r01 --output-format csv -- $B > gpurun_out/prof/trace_bench.log 2>&1
tail -2 gpurun_out/prof/trace_bench.log
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/pmc_fetch -o r01 --output-format csv -- $B > gpurun_out/prof/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/pmc_write -o r01 --output-format csv -- $B > gpurun_out/prof/pmc_write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/prof/pmc_sq -o r01 --output-format csv -- $B > gpurun_out/prof/pmc_sq.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/prof/pmc_sq2 -o r01 --output-format csv -- $B > gpurun_out/prof/pmc_sq2.log 2>&1
find gpurun_out/prof -type f | head -50
du -sh gpurun_out/prof

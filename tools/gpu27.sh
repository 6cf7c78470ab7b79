cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof6
B="python bench.py --steps 2 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/prof6/ic -o r --output-format csv -- $B > gpurun_out/prof6/ic.log 2>&1
tail -2 gpurun_out/prof6/ic.log
timeout -s KILL 600 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY --kernel-trace -d gpurun_out/prof6/lv -o r --output-format csv -- $B > gpurun_out/prof6/lv.log 2>&1
tail -2 gpurun_out/prof6/lv.log

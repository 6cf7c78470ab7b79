cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu 2>&1 | grep -E "value|Error|error" | cut -c1-300
timeout -s KILL 600 python bench.py --steps 4 --warmup 2 --no-cpu --concurrent 4096 2>&1 | grep -E "value|Error|error" | cut -c1-300
mkdir -p gpurun_out/prof5
B="python bench.py --steps 2 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/prof5/sq -o r --output-format csv -- $B > gpurun_out/prof5/sq.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/prof5/sq2 -o r --output-format csv -- $B > gpurun_out/prof5/sq2.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM --kernel-trace -d gpurun_out/prof5/sq3 -o r --output-format csv -- $B > gpurun_out/prof5/sq3.log 2>&1
tail -3 gpurun_out/prof5/sq3.log

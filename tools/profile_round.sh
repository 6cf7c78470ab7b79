#!/bin/bash
# Runs on the GPU box (via gpurun): collects the rocprofv3 evidence of the default bench command into
# gpurun_out/prof_round/.  tools/summarize_profiles.py turns it into profiles/rNN/ (tracked).
# PMC passes are separate from each other and use --kernel-trace only (no sys/hip/hsa trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu"
timeout -s KILL 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o r --output-format csv -- $B > $OUT/trace.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o r --output-format csv -- $B > $OUT/fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o r --output-format csv -- $B > $OUT/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --kernel-trace -d $OUT/tcp -o r --output-format csv -- $B > $OUT/tcp.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/sq -o r --output-format csv -- $B > $OUT/sq.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq2 -o r --output-format csv -- $B > $OUT/sq2.log 2>&1
find $OUT -name "*.csv" | head -30
du -sh $OUT
# ---- round 4 additions: L2 hit rate of the dense kernel, the whole search per kernel (capacity tiers), the layer-rebuilding engine
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d $OUT/tcc -o r --output-format csv -- $B > $OUT/tcc.log 2>&1
timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $OUT/proof_trace -o r --output-format csv -- python bench.py --cpu-seconds 1 > $OUT/proof_trace.log 2>&1
M="python bench.py --workload max2sat --instance frb15-9-1 --prove 10 --no-cpu"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/m2_trace -o r --output-format csv -- $M > $OUT/m2_trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/m2_fetch -o r --output-format csv -- $M > $OUT/m2_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/m2_write -o r --output-format csv -- $M > $OUT/m2_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/m2_sq -o r --output-format csv -- $M > $OUT/m2_sq.log 2>&1
find $OUT -name "*stats*.csv" | head; du -sh $OUT
# ---- round 6: FETCH_SIZE / WRITE_SIZE passes of the secondary workloads (bench.py --workload ...: roofline.traffic of those lines;
# tools/summarize_extra.py -> pmc_extra.json "secondary").  Every pass prints its own bench line: the nodes expanded under the counters.
sec() {   # sec <key> <bench.py arguments...>
  key=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    sub=${key}_$(echo $c | tr A-Z a-z | cut -d_ -f1)
    timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$sub -o r --output-format csv -- python bench.py "$@" --no-cpu > $OUT/$sub.log 2>&1
  done
}
sec max2sat_frb10_6_1 --workload max2sat
sec max2sat_frb15_9_1 --workload max2sat --instance frb15-9-1 --prove 10
sec mcp_n30 --workload mcp
sec tsptw_c5 --workload tsptw
du -sh $OUT

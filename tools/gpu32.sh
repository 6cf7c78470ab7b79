cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_ta_$1
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 2 --no-cpu"
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum --kernel-trace -d $OUT/a -o r --output-format csv -- $B > $OUT/a.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace -d $OUT/b -o r --output-format csv -- $B > $OUT/b.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --kernel-trace -d $OUT/c -o r --output-format csv -- $B > $OUT/c.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum TCP_TCC_UC_READ_REQ_sum --kernel-trace -d $OUT/d -o r --output-format csv -- $B > $OUT/d.log 2>&1
tail -2 $OUT/d.log | cut -c1-200

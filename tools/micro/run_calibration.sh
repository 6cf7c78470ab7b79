#!/bin/bash
# Runs on the GPU box (via gpurun): counter calibration kernels, plain (timings) and under rocprofv3 --pmc (one pass per counter).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
BIN=tools/micro/counter_calibration
timeout -s KILL 300 $BIN 512 512 64 4096 > $OUT/plain.jsonl 2> $OUT/plain.err
cat $OUT/plain.jsonl
timeout -s KILL 300 $BIN 2048 256 32 4096 > $OUT/plain_2048x256.jsonl 2>> $OUT/plain.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 600 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o r --output-format csv -- $BIN 512 512 64 4096 > $OUT/$c.log 2>&1
done
timeout -s KILL 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d $OUT/req -o r --output-format csv -- $BIN 512 512 64 4096 > $OUT/req.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $OUT/req2 -o r --output-format csv -- $BIN 512 512 64 4096 > $OUT/req2.log 2>&1
find $OUT -name "*.csv" | head; tail -3 $OUT/req.log $OUT/req2.log

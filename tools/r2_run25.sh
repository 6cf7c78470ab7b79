#!/bin/bash
# round 2, GPU run 25: rocprofv3 kernel statistics of (a) a time-budgeted whole search (all tiers: the small-DD kernels) and (b) config C3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run25; rm -rf $O; mkdir -p $O
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $O/search -o r --output-format csv -- python tools/search_stats.py brock400_1 10000 8192 30 > $O/search.log 2>&1; tail -2 $O/search.log | cut -c1-300
timeout -s KILL 600 rocprofv3 --kernel-trace --stats -d $O/c3 -o r --output-format csv -- python bench.py --workload max2sat --no-cpu > $O/c3.log 2>&1; tail -c 300 $O/c3.log
find $O -name "*kernel_stats.csv" | while read f; do echo $f; head -6 $f | cut -c1-200; done
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O

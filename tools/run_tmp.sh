# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout -s KILL 300 python bench.py --workload max2sat > gpurun_out/r05/bench_max2sat.json 2>/dev/null
timeout -s KILL 300 python bench.py --workload max2sat --instance frb15-9-1 --prove 30 > gpurun_out/r05/bench_max2sat_frb15-9-1.json 2>/dev/null
timeout -s KILL 300 python bench.py --workload mcp > gpurun_out/r05/bench_mcp.json 2>/dev/null
timeout -s KILL 300 python bench.py --workload tsptw > gpurun_out/r05/bench_tsptw.json 2>/dev/null
timeout -s KILL 300 python -m pytest tests/test_gpu_boundary_b1.py -q -s -m gpu -p no:cacheprovider 2>&1 | grep "T=\|passed\|failed" > gpurun_out/r05/boundary_b1_threads.txt
cat gpurun_out/r05/boundary_b1_threads.txt | cut -c1-250
for f in gpurun_out/r05/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('$f', 'value %.4e' % d['value'], 'frac %.4f' % r.get('frac',0), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'speedup', d.get('speedup_vs_cpu'))
"; done

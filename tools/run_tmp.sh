cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/lpt
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d.get('proof') or {}
    print(sys.argv[1].split('/')[-1], 'value %.4g'%d['value'], 'ms %.2f'%d['ms_per_step'], 'kernel_ms %.2f'%d['roofline']['kernel_ms_avg'], 'frac %.3f'%d['roofline']['frac'], 'proof', p.get('wall_s'), p.get('subproblems'), [round(t['kernel_s'],1) for t in p.get('tiers_rank0',[])])
except Exception as e:
    print(sys.argv[1],'ERR',e)
PY
}
for r in 1 2; do
  for l in 0 1; do
    DDO_HIP_LPT=$l timeout -s KILL 300 python bench.py --no-cpu > gpurun_out/lpt/b_${l}_$r.json 2> gpurun_out/lpt/b_${l}_$r.err; show gpurun_out/lpt/b_${l}_$r.json
  done
done
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api_surface.py tests/test_gpu_boundaries.py -x -q -p no:cacheprovider > gpurun_out/lpt/pytest.log 2>&1; tail -3 gpurun_out/lpt/pytest.log | cut -c1-300
for l in 1 0; do
  DDO_HIP_LPT=$l timeout -s KILL 400 python bench.py --cpu-seconds 2 > gpurun_out/lpt/p_$l.json 2> gpurun_out/lpt/p_$l.err; show gpurun_out/lpt/p_$l.json
done

cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
run() { env $2 timeout -s KILL 400 python bench.py --cpu-seconds 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['proof']; print('$1 proof %.2f s kernels %s handed_up %s' % (p['wall_s'], [round(t['kernel_s'],2) for t in p['tiers_rank0']], [t['handed_up'] for t in p['tiers_rank0']]))"; }
run "skip90(default)" "X=1"
run "skip50" "DDO_HIP_TIER_SKIP=50"
run "skip70" "DDO_HIP_TIER_SKIP=70"
run "skip30" "DDO_HIP_TIER_SKIP=30"
run "tiers 512:64,2048:128" "DDO_HIP_TIERS=512:64,2048:128"

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for B in 4096 8192; do python bench.py --no-cpu --concurrent $B | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($B, j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['kernel_nodes_per_s'])"; done

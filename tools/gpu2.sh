set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 300 python -c "
import torch
print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout -s KILL 900 python bench.py --steps 3 --warmup 2 --no-cpu 2>&1 | tail -5
timeout -s KILL 900 python bench.py --steps 3 --warmup 2 --no-cpu --concurrent 256 2>&1 | tail -3
for T in 1 32 128 256; do
timeout -s KILL 120 python - <<PY
import sys; sys.path.insert(0,'.')
from tests.oracle_binding import Oracle
o=Oracle('oracle/_build/liboracle.so'); i=o.misp('data/misp/brock400_1.clq')
r=i.solve(10000, $T, 12.0)
print('CPU T=$T', r['explored'], r['nodes_expanded'], r['wall_s'], r['nodes_expanded']/r['wall_s']/1e6, 'Mnodes/s', r['best_lb'], r['best_ub'])
PY
done

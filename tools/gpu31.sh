cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import ddo_amd
from ddo_amd import SubProblem
model = ddo_amd.Misp.read_instance("data/misp/brock400_1.clq")
W = int(sys.argv[1])
mdd = ddo_amd.Mdd(model, W)
state = np.zeros(model.ws, dtype=np.uint64)
for v in range(model.n): state[v // 64] |= np.uint64(1) << np.uint64(v % 64)
sub = SubProblem(state=state, value=0, path=[], depth=0)
comp = mdd.compile(2, W, sub, -10**9)
print(W, comp, flush=True)
PY
for cfg in "100 1024" "100 512" "2048 512" "2048 256" "2048 1024"; do set -- $cfg; echo "== W=$1 threads=$2"; DDO_HIP_THREADS=$2 timeout -s KILL 120 python /tmp/one.py $1 2>&1 | grep -v "^  File\|^$" | tail -3 | cut -c1-200; done

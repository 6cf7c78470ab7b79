cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DDO_HIP_STATS=1 timeout -s KILL 600 python bench.py --no-cpu > /tmp/o.json 2> /tmp/o.err
python - <<'PY'
import json
j=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['launches'], j['roofline']['kernel_nodes_per_s'])
PY
grep "host s\|arena download" /tmp/o.err | sed 's/.*host s/host s/'

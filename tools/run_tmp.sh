cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mv ddo_amd/_build ddo_amd/_build_base
for v in _build_base _build_prev _build_base; do
rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
for w in mcp tsptw max2sat; do timeout -s KILL 300 python bench.py --workload $w --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v $w kernel_s %.4f wall_s %.4f' % (d['roofline']['kernel_s'], d['roofline']['wall_s']))"; done
timeout -s KILL 300 python bench.py --workload tsptw --instance AFG/rbg125a.tw --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v rbg125a kernel_s %.4f wall_s %.4f' % (d['roofline']['kernel_s'], d['roofline']['wall_s']))"
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build
timeout -s KILL 200 python bench.py --workload max2sat --instance frb15-9-1 --prove 20 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('frb15 %.4g' % d['value'])"

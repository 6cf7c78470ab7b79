cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/ab_frb15.sh _build_prev 2>&1 | tail -4
mv ddo_amd/_build ddo_amd/_build_base
for v in _build_base _build_prev; do
rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
for w in max2sat mcp tsptw; do timeout -s KILL 300 python bench.py --workload $w --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v $w value %.4g kernel_s %.4f wall_s %.4f' % (d['value'], d['roofline']['kernel_s'], d['roofline']['wall_s']))"; done
timeout -s KILL 300 python bench.py --workload tsptw --instance AFG/rbg125a.tw --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v rbg125a value %.4g kernel_s %.4f wall_s %.4f' % (d['value'], d['roofline']['kernel_s'], d['roofline']['wall_s']))"
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_base ddo_amd/_build

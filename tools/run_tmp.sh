cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for i in 1; do timeout -s KILL 600 python -m pytest tests/test_gpu_api_surface.py -x -q -p no:cacheprovider -k launch_order 2>&1 | tail -1; done
timeout -s KILL 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -1 gpurun_out/profile_round.log
echo finished

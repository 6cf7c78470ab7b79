#!/bin/bash
# Runs on the GPU box: quick parity subset of the in-place kernels, then the frozen bench three times.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/micro
if [ -x tools/micro/expand_patterns ] && [ ! -s gpurun_out/micro/expand_patterns_512.txt ]; then
  ./tools/micro/expand_patterns 512 10000 > gpurun_out/micro/expand_patterns_512.txt 2>&1
  ./tools/micro/expand_patterns 256 10000 > gpurun_out/micro/expand_patterns_256.txt 2>&1
  cat gpurun_out/micro/expand_patterns_512.txt
fi
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or shrunk or maximum or (replay_of_oracle and (brock200_2 or keller4))" 2>&1 | tail -4
for rep in 1 2 3; do
  timeout -s KILL 300 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g ms/step %.2f frac %.4f kernel_ms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg']))"
done

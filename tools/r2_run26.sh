#!/bin/bash
# round 2, GPU run 26: next tier launched before the finished tier's results are decoded -- parity suites, time to proof A/B of two builds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run26; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_api_surface.py -m gpu -q -x 2>&1 | tail -3
mv ddo_amd/_build ddo_amd/_build_new
for rep in 1 2; do
 for v in _build_old _build_new; do
   rm -rf ddo_amd/_build; cp -r ddo_amd/$v ddo_amd/_build
   echo "$v: $(DDO_HIP_STATS=1 timeout 400 python tools/search_stats.py brock400_1 10000 8192 300 2> $O/err_${v}_$rep.txt | sed 's/{[^}]*}//')" | tee -a $O/proof_ab.txt
   grep "host s:" $O/err_${v}_$rep.txt | cut -c1-160 | tee -a $O/proof_ab.txt
 done
done
rm -rf ddo_amd/_build; mv ddo_amd/_build_new ddo_amd/_build

# scratch: the command of the last `gpurun -- 'bash tools/run_tmp.sh'` of a session (rewritten before every call; nothing depends on it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "== LEX_CAP=3"; DDO_HIP_LEX_CAP=3 timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pooled.py -x -q -m gpu -k "golden or replay" -p no:cacheprovider 2>&1 | tail -2
echo "== ENGINE=1"; DDO_HIP_ENGINE=1 timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "(golden or replay or sequential_parity) and not dense and not tier and not auto" 2>&1 | tail -2
for cfg in DDO_HIP_THREADS=512 DDO_HIP_KEYS_GLOBAL=1 DDO_HIP_SLOTS=7 DDO_HIP_SPLIT=0 DDO_HIP_NO_AUTO_DENSE=1; do
    echo "== $cfg"; env $cfg timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary_b1.py tests/test_gpu_shim.py -x -q -m gpu -p no:cacheprovider -k "golden or replay or lazy or concurrent or reference" 2>&1 | tail -1
done

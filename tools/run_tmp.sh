cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gpu_api_surface.py -x -q -m gpu 2>&1 | tail -3
timeout -s KILL 600 python bench.py --cpu-seconds 4 > gpurun_out/bench_kp.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_kp.json").read().strip().splitlines()[-1])
p=d.get("proof",{})
print("value %.4g frac %.4f" % (d["value"], d["roofline"]["frac"]), "proof_s", p.get("time_to_proved_optimum_s"), "kernels", p.get("kernel_s"), [ (t.get("kernel_ms"), t.get("subproblems"), t.get("handed_up")) for t in p.get("tiers_rank0",[])])
PY

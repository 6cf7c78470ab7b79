#!/usr/bin/env python3
"""Turns gpurun_out/prof_round (tools/profile_round.sh) into profiles/<round>/: the rocprofv3 kernel statistics, the
per-dispatch PMC counters of the compile kernel and pmc_summary.json (what bench.py reports as roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01"
os.makedirs(dst, exist_ok=True)
KERNEL = "misp_compile_kernel"


def find(sub, suffix):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(root, f)
    return None


def kernel_dispatches(sub):
    """dispatch id -> (kernel name, duration ns) for the compile kernel, in launch order"""
    out = collections.OrderedDict()
    p = find(sub, "kernel_trace.csv")
    for r in csv.DictReader(open(p)):
        if KERNEL in r["Kernel_Name"]:
            out[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def counters(sub):
    p = find(sub, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(p)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
steps, warm = bench["steps"], bench["warmup"]
summary = {"bench": {k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "roofline": bench["roofline"]}
# what these counters describe: bench.py refuses to quote them for another build or another dominant kernel
try:
    import subprocess
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "ddo_amd/csrc"], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip()
    git = git + ("+uncommitted" if dirty else "")
except OSError:
    git = None
summary["stamp"] = {"kernel_sources": bench["roofline"].get("kernel_sources"), "kernel": bench["roofline"].get("kernel"), "git": git,
                    "command": "python bench.py (tools/profile_round.sh: kernel trace + one rocprofv3 --pmc pass per counter group)"}

for name in ("kernel_stats.csv", "kernel_trace.csv"):
    p = find("trace", name)
    if p:
        shutil.copy(p, os.path.join(dst, name))
disp = kernel_dispatches("trace")
durs = [d for _, d in disp.values()]
timed = durs[-steps:]
summary["kernel_trace"] = {"launches_ms": [d / 1e6 for d in durs], "timed_avg_ms": sum(timed) / len(timed) / 1e6,
                           "note": "root sub-problem (1 workgroup), warm-up launches, then the timed launches"}

rows = []
for sub in ("fetch", "write", "tcp", "sq", "sq2"):
    if not find(sub, "counter_collection.csv"):
        continue
    d = kernel_dispatches(sub)
    c = counters(sub)
    ids = list(d.keys())
    for i, k in enumerate(ids):
        for cn, v in sorted(c[k].items()):
            rows.append({"pass": sub, "launch": i, "duration_ms": d[k][1] / 1e6, "counter": cn, "value": v})
    tids = ids[-steps:]
    for cn in sorted({cn for k in tids for cn in c[k]}):
        summary.setdefault("pmc_timed_avg", {})[cn] = sum(c[k][cn] for k in tids) / len(tids)
    summary.setdefault("pmc_timed_ms", {})[sub] = sum(d[k][1] for k in tids) / len(tids) / 1e6
with open(os.path.join(dst, "pmc_counters.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["pass", "launch", "duration_ms", "counter", "value"])
    w.writeheader()
    w.writerows(rows)

pa = summary.get("pmc_timed_avg", {})
if "FETCH_SIZE" in pa and "WRITE_SIZE" in pa:
    # FETCH_SIZE / WRITE_SIZE are in units of 1 KB.  What a counter byte is worth depends on the access pattern; it was
    # calibrated on this kernel's patterns with known byte counts (tools/micro/counter_calibration.hip ->
    # profiles/<round>/counter_calibration.json): a 64-byte record line read by one lane (4 x 16 B) or 8 bytes of it (the
    # work-list sweep) within a DD's slot count 1 / 1.10 and 1 / 1.14 of their lines -- NOT the 1/2 the guide measured for wide
    # coalesced streams, which this kernel has next to none of -- and WRITE_SIZE counts a whole line as 64 B (x 1.07) and an
    # 8-byte store as the 32-byte sector it moves.  One figure: FETCH_SIZE x fetch_factor + WRITE_SIZE.
    cal = None
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (os.path.join(dst, "counter_calibration.json"), os.path.join(here, "..", "profiles", "r04", "counter_calibration.json")):
        if os.path.exists(cand):
            cal = json.load(open(cand))["patterns"]
            break
    ff = 0.5 * (cal["read_line64_slot"]["fetch_factor_lines"] + cal["read_word8_slot"]["fetch_factor_lines"]) if cal else 2.0
    fetch = pa["FETCH_SIZE"] * 1024.0
    write = pa["WRITE_SIZE"] * 1024.0
    summary["traffic"] = {"fetch_bytes_raw": fetch, "fetch_factor": ff, "write_bytes": write,
                          "fetch_factor_source": "counter_calibration.json: mean of read_line64_slot and read_word8_slot" if cal else "guide's x2 (no calibration file)",
                          "hbm_bytes_per_launch": ff * fetch + write,
                          "hbm_bytes_per_node": (ff * fetch + write) / bench["roofline"]["nodes_per_launch"],
                          "algorithmic_bytes_per_launch": bench["roofline"]["nodes_per_launch"] * bench["roofline"]["bytes_per_node"]}
    summary["traffic"]["ratio_to_algorithmic"] = summary["traffic"]["hbm_bytes_per_launch"] / summary["traffic"]["algorithmic_bytes_per_launch"]
    # the bench run of a collection precedes its own counter passes: its line cannot carry them yet; this copy does
    summary["roofline"]["traffic"] = ff * fetch + write
    summary["roofline"]["traffic_source"] = "this collection's rocprofv3 --pmc passes (FETCH_SIZE x %.2f [calibrated] + WRITE_SIZE, timed launches)" % ff
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])

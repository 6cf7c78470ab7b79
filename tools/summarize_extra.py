#!/usr/bin/env python3
"""Round-4 additions of tools/profile_round.sh -> profiles/<round>/pmc_extra.json: L2 hit rate of the dense kernel (timed launches),
the whole search per kernel (rocprofv3 --stats of a bench run with the proof), the layer-rebuilding engine on MAX2SAT frb15-9-1
(kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes).   python tools/summarize_extra.py gpurun_out/prof_round profiles/r05"""
import collections
import csv
import json
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_round"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05"


def agg(sub, key):
    a = collections.OrderedDict()
    for r in csv.DictReader(open(f"{src}/{sub}/r_counter_collection.csv")):
        if key in r["Kernel_Name"]:
            a.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    return a


def durs(sub, key):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f"{src}/{sub}/r_kernel_trace.csv")):
        if key in r["Kernel_Name"]:
            d[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


out = {}
a = agg("tcc", "dense")
ids = sorted(a)[-8:]
t = collections.defaultdict(float)
for i in ids:
    for k, v in a[i].items():
        t[k] += v / len(ids)
t = dict(t)
t["l2_hit_rate"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
out["dense_kernel_l2"] = t
shutil.copy(f"{src}/proof_trace/r_kernel_stats.csv", f"{dst}/proof_kernel_stats.csv")
rows = list(csv.DictReader(open(f"{src}/proof_trace/r_kernel_stats.csv")))
out["whole_search_kernels"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in rows[:4]]
shutil.copy(f"{src}/m2_trace/r_kernel_stats.csv", f"{dst}/max2sat_frb15_kernel_stats.csv")
m = {}
for sub in ("m2_fetch", "m2_write", "m2_sq"):
    a = agg(sub, "misp_compile_kernel")
    d = durs(sub, "misp_compile_kernel")
    tot = collections.defaultdict(float)
    for i in a:
        for k, v in a[i].items():
            tot[k] += v
    m[sub] = {"launches": len(a), "kernel_s": sum(d.values()) / 1e9, **dict(tot)}
out["max2sat_frb15_9_1"] = m
bm = [json.loads(l) for l in open(f"{src}/m2_trace.log") if l.startswith("{")]
if bm:
    b = bm[-1]
    out["max2sat_frb15_9_1"]["bench_under_trace"] = {"value": b["value"], "roofline": {k: b["roofline"][k] for k in ("achieved", "frac", "kernel_s", "bytes_per_node", "kernel_nodes_per_s")}}
# ---- round 6: the secondary workloads' HBM traffic per expanded node (bench.py: committed_traffic).  Each counter pass printed its
# own bench line; `nodes_all_passes` = the nodes expanded by the process the counters were wrapped around (warm-up pass included).
import os

cal = None
for cand in (f"{dst}/counter_calibration.json", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r05", "counter_calibration.json"),
             os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r04", "counter_calibration.json")):
    if os.path.exists(cand):
        cal = json.load(open(cand))["patterns"]
        break
ff = 0.5 * (cal["read_line64_slot"]["fetch_factor_lines"] + cal["read_word8_slot"]["fetch_factor_lines"]) if cal else 2.0
sec = {}
for key in ("max2sat_frb10_6_1", "max2sat_frb15_9_1", "mcp_n30", "tsptw_c5"):
    ent = {}
    try:
        for cname, sub in (("FETCH_SIZE", key + "_fetch"), ("WRITE_SIZE", key + "_write")):
            a = agg(sub, "misp_compile_kernel")
            d = durs(sub, "misp_compile_kernel")
            line = [json.loads(l) for l in open(f"{src}/{sub}.log") if l.startswith("{")][-1]
            kb = sum(v[cname] for v in a.values())
            ent[cname] = {"kb": kb, "launches": len(a), "kernel_s": sum(d.values()) / 1e9, "nodes": line["nodes_all_passes"],
                          "bytes_per_node_raw": kb * 1024.0 / line["nodes_all_passes"]}
            ent["kernel_sources"] = line["kernel_sources"]
            ent["algorithmic_bytes_per_node"] = line["roofline"]["bytes_per_node"]
            ent["roofline_frac_under_counters"] = line["roofline"]["frac"]
        ent["fetch_factor"] = ff
        ent["nodes"] = ent["FETCH_SIZE"]["nodes"]
        ent["hbm_bytes_per_node"] = ff * ent["FETCH_SIZE"]["bytes_per_node_raw"] + ent["WRITE_SIZE"]["bytes_per_node_raw"]
        ent["ratio_to_algorithmic"] = ent["hbm_bytes_per_node"] / ent["algorithmic_bytes_per_node"]
        sec[key] = ent
    except (OSError, IndexError, KeyError, ZeroDivisionError) as e:
        sec[key] = {"error": repr(e)}
out["secondary"] = sec
json.dump(out, open(f"{dst}/pmc_extra.json", "w"), indent=1)
print(json.dumps({k: {q: v.get(q) for q in ("hbm_bytes_per_node", "ratio_to_algorithmic", "error")} for k, v in sec.items()}))
print(json.dumps(out["dense_kernel_l2"]), out["whole_search_kernels"][:2])

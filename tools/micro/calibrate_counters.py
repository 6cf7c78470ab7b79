#!/usr/bin/env python3
"""Joins the JSON lines of tools/micro/counter_calibration (known byte counts) with the rocprofv3 --pmc passes that
tools/micro/run_calibration.sh collected and writes profiles/<round>/counter_calibration.json: for every access pattern the
factor  true bytes / counter bytes  of FETCH_SIZE and WRITE_SIZE (1 KB units), and the throughput the pattern sustains.
    python tools/micro/calibrate_counters.py gpurun_out/calib profiles/r04"""
import collections
import csv
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/calib"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04"
cases = [json.loads(l) for l in open(os.path.join(src, "plain.jsonl")) if l.startswith("{") and "skipped" not in l]


def per_dispatch(sub):
    """kernel dispatches in launch order -> {counter: value summed over XCDs / instances}"""
    p = os.path.join(src, sub, "r_counter_collection.csv")
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        if "k_read" in r["Kernel_Name"] or "k_write" in r["Kernel_Name"]:
            agg.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    return [agg[k] for k in sorted(agg)]


out = {"source": "tools/micro/counter_calibration 512 512 64 4096 (512 workgroups x 512 threads, 4 GB buffer; 'slot' = random lines within a "
                 "14 352-line region per workgroup, 'far' = random lines over the whole buffer); second launch of every case",
       "units": "FETCH_SIZE / WRITE_SIZE are reported in units of 1 KB", "patterns": {}}
passes = {s: per_dispatch(s) for s in ("FETCH_SIZE", "WRITE_SIZE", "req", "req2") if os.path.exists(os.path.join(src, s, "r_counter_collection.csv"))}
for i, c in enumerate(cases):
    row = {k: c[k] for k in ("useful_bytes", "line_bytes", "ms", "useful_GBps", "line_GBps", "Mlines_per_s")}
    for s, d in passes.items():
        if len(d) != 2 * len(cases):
            continue
        for cn, v in d[2 * i + 1].items():   # the second launch of the case
            row[cn] = v
    if "FETCH_SIZE" in row and not c["write"]:
        row["fetch_bytes_counter"] = row["FETCH_SIZE"] * 1024.0
        row["fetch_factor_useful"] = c["useful_bytes"] / row["fetch_bytes_counter"]
        row["fetch_factor_lines"] = c["line_bytes"] / row["fetch_bytes_counter"]
    if "WRITE_SIZE" in row and c["write"]:
        row["write_bytes_counter"] = row["WRITE_SIZE"] * 1024.0
        row["write_factor_useful"] = c["useful_bytes"] / row["write_bytes_counter"]
        row["write_factor_lines"] = c["line_bytes"] / row["write_bytes_counter"]
    out["patterns"][c["case"]] = row
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "counter_calibration.json"), "w"), indent=1)
for k, r in out["patterns"].items():
    print("%-20s %8.1f GB/s lines  fetchF(lines) %-6s writeF(lines) %-6s  TCP_RD %-12s TCP_WR %-12s EA_RD %-12s EA_WR %-12s hit %-10s miss %-10s" % (
        k, r["line_GBps"], "%.2f" % r["fetch_factor_lines"] if "fetch_factor_lines" in r else "-", "%.2f" % r["write_factor_lines"] if "write_factor_lines" in r else "-",
        r.get("TCP_TCC_READ_REQ_sum", "-"), r.get("TCP_TCC_WRITE_REQ_sum", "-"), r.get("TCC_EA0_RDREQ_sum", "-"), r.get("TCC_EA0_WRREQ_sum", "-"), r.get("TCC_HIT_sum", "-"), r.get("TCC_MISS_sum", "-")))

#!/usr/bin/env python3
"""Replaces ONE workload's entry of profiles/<tag>_hbm_traffic.json by the counter passes that
tools/pmc_traffic_one.sh left under gpurun_out/prof_<tag>/ (same arithmetic as
tools/summarize_prof.py).  Usage: python tools/merge_traffic.py c4 r03"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, tag = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
per = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    hits = glob.glob(os.path.join(src, f"pmc_{wl}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        sys.exit(f"no {ctr} pass for {wl}")
    with open(hits[0], newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != ctr:
                continue
            m = re.search(r"([A-Za-z0-9_]+_kernel)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"]
            s = per.setdefault(k, {}).setdefault(ctr, [0.0, 0])
            s[0] += float(row["Counter_Value"])
            s[1] += 1
out = {}
for k, e in per.items():
    f, w = e.get("FETCH_SIZE", [0.0, 0]), e.get("WRITE_SIZE", [0.0, 0])
    fk = f[0] / f[1] if f[1] else None
    wk = w[0] / w[1] if w[1] else None
    rec = {"dispatches": max(f[1], w[1]), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk}
    if fk is not None and wk is not None:
        rec["hbm_bytes"] = round((2 * fk + wk) * 1024)
        rec["hbm_bytes_uncorrected"] = round((fk + wk) * 1024)
    out[k] = rec
path = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")
traffic = json.load(open(path))
traffic[wl] = out
with open(path, "w") as fh:
    json.dump(traffic, fh, indent=1, sort_keys=True)
print(json.dumps({k: v.get("hbm_bytes") for k, v in out.items()}, indent=1))

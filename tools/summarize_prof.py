#!/usr/bin/env python3
"""Condenses what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the files kept
under profiles/:

  profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_hbm_traffic.json              per kernel and workload: mean FETCH_SIZE and
                                               WRITE_SIZE per dispatch from the --pmc passes

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte read request of
a wide coalesced stream as 64 bytes, so `hbm_bytes` below is (2 * FETCH_SIZE + WRITE_SIZE) *
1024 and `hbm_bytes_uncorrected` is (FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_name(kernel):
    m = re.search(r"([A-Za-z0-9_]+_kernel)", kernel)
    return m.group(1) if m else kernel


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r00"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for wl in ("framepair", "c3", "c4", "c5", "f1", "f2", "f3", "vo"):
        hits = glob.glob(os.path.join(src, f"stats_{wl}", "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"))
            print("kept", hits[0])
    traffic = {}
    for wl in ("framepair", "c3", "c4"):
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            hits = glob.glob(os.path.join(src, f"pmc_{wl}_{ctr}", "**", "*counter_collection.csv"),
                             recursive=True)
            if not hits:
                continue
            with open(hits[0], newline="") as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") != ctr:
                        continue
                    k = short_name(row["Kernel_Name"])
                    e = per.setdefault(k, {})
                    s = e.setdefault(ctr, [0.0, 0])
                    s[0] += float(row["Counter_Value"])
                    s[1] += 1
        out = {}
        for k, e in per.items():
            f = e.get("FETCH_SIZE", [0.0, 0])
            w = e.get("WRITE_SIZE", [0.0, 0])
            fk = f[0] / f[1] if f[1] else None
            wk = w[0] / w[1] if w[1] else None
            rec = {"dispatches": max(f[1], w[1]), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk}
            if fk is not None and wk is not None:
                rec["hbm_bytes"] = round((2 * fk + wk) * 1024)
                rec["hbm_bytes_uncorrected"] = round((fk + wk) * 1024)
            out[k] = rec
        if out:
            traffic[wl] = out
    if traffic:
        path = os.path.join(dst, f"{tag}_hbm_traffic.json")
        with open(path, "w") as f:
            json.dump(traffic, f, indent=1, sort_keys=True)
        print("wrote", path)
        print(json.dumps(traffic, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

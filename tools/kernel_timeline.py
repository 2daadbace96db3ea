#!/usr/bin/env python3
"""Prints the kernels of the last occurrence of a step in a rocprofv3 kernel trace:
   kernel_timeline.py <trace.csv> <first-kernel-substring> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + int(sys.argv[3]) if len(sys.argv) > 3 else idx + 16]:
    n = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][-40:]
    print(f"{n:40s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur "
          f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  grid "
          f"{r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")

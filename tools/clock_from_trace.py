#!/usr/bin/env python3
"""Shader clock per kernel from one rocprofv3 run with --kernel-trace --pmc GRBM_GUI_ACTIVE:
clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs, so / 8) / dispatch duration.  usage: clock_from_trace.py <dir>"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cyc = collections.defaultdict(float)
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
agg = collections.OrderedDict()
for k, (name, ns) in sorted(dur.items()):
    if "ocr::" not in name or k not in cyc:
        continue
    short = name.split("(")[0].replace("void ocr::igemm::", "").replace("ocr::(anonymous namespace)::", "")
    e = agg.setdefault(short, [0.0, 0.0, 0])
    e[0] += cyc[k] / 8.0
    e[1] += ns
    e[2] += 1
print(f"{'kernel':56s} {'launches':>8s} {'avg us':>9s} {'GHz':>6s}")
for name, (c, ns, n) in agg.items():
    print(f"{name[:56]:56s} {n:8d} {ns / n / 1e3:9.1f} {c / ns:6.3f}")

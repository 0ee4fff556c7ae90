#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files: pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

rows = collections.OrderedDict()
counters = []
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ocr::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ocr::igemm::", "").replace("ocr::", "")
            c = r["Counter_Name"]
            if c not in counters:
                counters.append(c)
            e = rows.setdefault(name, collections.defaultdict(float))
            e[c] += float(r["Counter_Value"])
            e["_n_" + c] += 1
print(f"{'kernel':58s} " + " ".join(f"{c[-22:]:>22s}" for c in counters))
for name, e in rows.items():
    print(f"{name[:58]:58s} " + " ".join(f"{e[c]:22.4g}" for c in counters))

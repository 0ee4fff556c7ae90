#!/usr/bin/env python3
"""Builds profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in
SEPARATE runs of tools/profile_layers.py 32 640 1, as MI355X_MICROARCH.md prescribes: the TCC block
cannot hold both).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a wide coalesced stream, so it is doubled.  usage: make_traffic_json.py <fetch_dir> <write_dir>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = {"128, 128": "128x128", "128, 64": "128x64", "64, 64": "64x64"}


def pretty(name: str) -> str:
    # "void ocr::igemm::conv_igemm<float, float, 64, 64, 3, 1, 0, 0>(...)" -> bench.py's kernel label
    if "conv_igemm<" not in name:
        return "stem_conv7x7_bn_relu_maxpool" if "stem_kernel" in name else \
               "convt2x2_sigmoid" if "convt2_sigmoid" in name else \
               "tail_convt1_bn_relu_convt2_sigmoid" if "tail_fused" in name else \
               "winograd_fused<c64>" if "winograd_fused_kernel<2>" in name else \
               "winograd_fused<c128>" if "winograd_fused_kernel<4>" in name else \
               "winograd_fused<c256>" if "winograd_fused_kernel<8>" in name else \
               "winograd_input_transform" if "winograd_input" in name else \
               "winograd_output_transform" if "winograd_output" in name else name.split("(")[0]
    a = [v.strip() for v in name.split("<")[1].split(">")[0].split(",")]
    ty = "bf16" if "bf16" in a[0] or "__bf16" in a[0] else "f32"
    tile = f"{a[2]}x{a[3]}"
    store = {"0": "", "1": ",SHUFFLE2", "2": ",PHASE"}[a[7]]
    src = {"0": "PLAIN", "2": "CAT4", "3": "PYR4"}[a[6]]
    return f"conv_igemm_{ty}<{tile},k{a[4]},s{a[5]},{src}{store}>"


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "ocr::" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])] = (pretty(r["Kernel_Name"]), per.get(int(r["Dispatch_Id"]), ("", 0.0))[1] + float(r["Counter_Value"]))
    return [per[k] for k in sorted(per)]


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
half = len(fetch) // 2                     # two identical forwards per run: keep the second (warm) one
agg = collections.OrderedDict()
for (nf, f), (nw, w) in zip(fetch[half:], write[half:]):
    assert nf == nw, (nf, nw)
    e = agg.setdefault(nf, [0.0, 0.0, 0])
    e[0] += f * 1024 * 2                   # KiB -> bytes, gfx950 half-count correction
    e[1] += w * 1024
    e[2] += 1
out = {"batch": 32, "size": 640, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 "
       "(gfx950), KiB units; profiles/r01_pmc_fetch_write.csv", "kernels": {}}
with open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write.csv"), "w") as fcsv:
    fcsv.write("kernel,launches,fetch_bytes_per_launch_corrected,write_bytes_per_launch\n")
    for k, (f, w, n) in agg.items():
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(f / n), "write_bytes_per_launch": round(w / n),
                             "bytes_per_launch": round((f + w) / n)}
        fcsv.write(f'"{k}",{n},{f / n:.0f},{w / n:.0f}\n')
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))

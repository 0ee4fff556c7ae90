#!/usr/bin/env python3
"""Builds profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in
SEPARATE runs of tools/profile_layers.py 32 640 1, as MI355X_MICROARCH.md prescribes: the TCC block
cannot hold both).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a wide coalesced stream, so it is doubled.  usage: make_traffic_json.py <fetch_dir> <write_dir> [round tag, default r02]
The extract is stamped with the git hash of the tree it was taken on, so bench.py's `roofline.traffic_source` shows a
stale file for what it is."""
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
        return "stem_x3_conv7x7_bn_relu_maxpool" if "stem_bf16_kernel<true" in name else \
               "stem_conv7x7_bn_relu_maxpool" if ("stem_kernel" in name or "stem_bf16_kernel" in name) else \
               "convt2x2_sigmoid" if "convt2_sigmoid" in name else \
               "tail_x3_convt1_bn_relu_convt2_sigmoid" if "tail_fused_kernel<float, true>" in name else \
               "tail_convt1_bn_relu_convt2_sigmoid" if "tail_fused" in name else \
               "rec_conv<4>" if "rec_conv_kernel<4>" in name else "rec_conv<2>" if "rec_conv_kernel<2>" in name else \
               "rec_conv<1>" if "rec_conv_kernel<1>" in name else "rec_fc2_softmax_top1" if "rec_fc2_softmax" in name else \
               "winograd43_fused<c64>" if "winograd43_fused_kernel<4>" in name else \
               "winograd43_fused<c128>" if "winograd43_fused_kernel<8>" in name else \
               "winograd43_fused<c256>" if "winograd43_fused_kernel<16>" in name else \
               "winograd43_input_transform" if "winograd43_input" in name else \
               "winograd43_output_transform" if "winograd43_output" in name else \
               "rec_small_fused" if "rec_small_fused" in name else \
               "rec_conv_small" if "rec_conv_small" in name else "rec_fc1_small" if "rec_fc1_small" in name else \
               "winograd_input_transform" if "winograd_input" in name else \
               "winograd_output_transform" if "winograd_output" in name else name.split("(")[0]
    a = [v.strip() for v in name.split("<")[1].split(">")[0].split(",")]
    ty = "x3" if (len(a) > 8 and a[8] == "true") else "bf16" if "bf16" in a[0] or "__bf16" in a[0] else "f32"
    tile = f"{a[2]}x{a[3]}"
    store = {"0": "", "1": ",SHUFFLE2", "2": ",PHASE"}[a[7]]
    if a[7] == "2":   # bench.py's label carries the upsampling factor; the 3x3 PYR4 form is up 8, the 2x2 forms are told apart by dispatch order
        store = ",PHASE8" if a[6] == "3" else ",PHASE2"
    src = {"0": "PLAIN", "2": "CAT4", "3": "PYR4"}[a[6]]
    return f"conv_igemm_{ty}<{tile},k{a[4]},s{a[5]},{src}{store}>"


def rows_of(d, counter):
    """(dispatch id, kernel name, value) of one counter from a rocprofv3 output directory: the rocpd SQLite database
    (ROCm 7.2 default) or the older *_counter_collection.csv."""
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if dbs:
        import sqlite3
        c = sqlite3.connect(dbs[0])
        for did, name, val in c.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)):
            yield int(did), name, float(val)
        return
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            yield int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])


def load(d, counter):
    per = collections.OrderedDict()
    for did, name, val in rows_of(d, counter):
        if "ocr::" in name:
            per[did] = (pretty(name), per.get(did, ("", 0.0))[1] + val)
    return [per[k] for k in sorted(per)]


args = [v for v in sys.argv[1:] if not v.startswith("--")]
fetch = load(args[0], "FETCH_SIZE")
write = load(args[1], "WRITE_SIZE")
half = len(fetch) // 2                     # two identical forwards per run: keep the second (warm) one
agg = collections.OrderedDict()
for (nf, f), (nw, w) in zip(fetch[half:], write[half:]):
    assert nf == nw, (nf, nw)
    e = agg.setdefault(nf, [0.0, 0.0, 0])
    e[0] += f * 1024 * 2                   # KiB -> bytes, gfx950 half-count correction
    e[1] += w * 1024
    e[2] += 1
import subprocess
tag = args[2] if len(args) > 2 else "r02"
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "ocr-rs_amd/csrc"], capture_output=True, text=True).stdout.strip()
out = {"batch": 32, "size": 640, "git_head": head + ("+dirty" if dirty else ""),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 "
       f"(gfx950), KiB units; profiles/{tag}_pmc_fetch_write.csv", "kernels": {}}
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_write.csv"), "w") as fcsv:
    fcsv.write("kernel,launches,fetch_bytes_per_launch_corrected,write_bytes_per_launch\n")
    for k, (f, w, n) in agg.items():
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(f / n), "write_bytes_per_launch": round(w / n),
                             "bytes_per_launch": round((f + w) / n)}
        fcsv.write(f'"{k}",{n},{f / n:.0f},{w / n:.0f}\n')
if len(args) > 4:   # recogniser passes (tools/profile_rec.py 65536): fetch dir, write dir
    rf, rw = load(args[3], "FETCH_SIZE"), load(args[4], "WRITE_SIZE")
    h2 = len(rf) // 2
    ragg = collections.OrderedDict()
    for (nf, f), (nw, w) in zip(rf[h2:], rw[h2:]):
        assert nf == nw, (nf, nw)
        nm = "rec_fc1" if nf.startswith("conv_igemm") else nf
        e = ragg.setdefault(nm, [0.0, 0.0, 0])
        e[0] += f * 1024 * 2
        e[1] += w * 1024
        e[2] += 1
    out["recogniser_b65536"] = {k: {"launches": n, "fetch_bytes_per_launch": round(f / n), "write_bytes_per_launch": round(w / n),
                                    "bytes_per_launch": round((f + w) / n)} for k, (f, w, n) in ragg.items()}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))

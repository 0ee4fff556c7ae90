#!/usr/bin/env python3
"""Builds profiles/pmc.json (what bench.py's roofline block cites) from rocprofv3 PMC passes over
tools/profile_layers.py 32 640 1 - each counter set in its OWN pass, as MI355X_MICROARCH.md prescribes (the TCC block cannot
hold FETCH_SIZE and WRITE_SIZE together; no tracing besides --kernel-trace in a counter pass):
    fetch dir : --pmc FETCH_SIZE                                   KiB; on gfx950 half of the bytes of a wide coalesced stream -> x 2
    write dir : --pmc WRITE_SIZE                                   KiB
    mops dir  : --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16      x 512 = MFMA FLOPs
    busy dir  : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE     busy share = BUSY / (1024 SIMDs x GUI_ACTIVE / 8 XCDs)
usage: make_pmc_json.py <f32|bf16> <fetch_dir> <write_dir> <mops_dir> <busy_dir> [round tag, default r04]
       [<rec_fetch_dir> <rec_write_dir>] [--labels <layers table of tools/profile_layers.py for the same options>]
--labels: name every dispatch of the forward by its position in the engine's launch order (one kernel per table row) instead of
by its template arguments alone - several launch kinds share one instantiation (the batched Winograd GEMMs and the 1x1 laterals).
Per kernel label (bench.py's names) and per launch, second (warm) forward of each run.  The extract is stamped with a hash
of ocr-rs_amd/csrc (bench.csrc_hash): bench.py marks it stale when the kernel sources change afterwards."""
import collections
import re
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = {"128, 128": "128x128", "128, 64": "128x64", "64, 64": "64x64"}


def pretty(name: str) -> str:
    # "void ocr::igemm::conv_igemm<float, float, 64, 64, 3, 1, 0, 0>(...)" -> bench.py's kernel label
    m = re.search(r"conv_igemmI(DF16b|f)(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
    if m:   # a name rocprofv3 left mangled (__bf16 template arguments)
        name = "void ocr::igemm::conv_igemm<%s, %s, %s, %s, %s, %s, %s, %s, %s>(" % (
            "__bf16" if m.group(1) != "f" else "float", "__bf16" if m.group(2) != "f" else "float", m.group(3), m.group(4), m.group(5),
            m.group(6), m.group(7), m.group(8), "true" if m.group(9) == "1" else "false")
    if "conv3x3_bf16_c64" in name:
        return "conv3x3_bf16_c64"
    if "conv_igemm<" not in name:
        return "stem_x3_conv7x7_bn_relu_maxpool" if ("stem_bf16_kernel<true" in name or "stem_bf16_kernelILb1E" in name) else \
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
    if ty == "x3" and a[4] == "1" and not store:   # the only 1x1 split-bf16 launches of the detector are the Winograd GEMMs
        store = ",BATCHED"
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
    """[(label, counter value summed over its rows)] per dispatch, in dispatch order"""
    per = collections.OrderedDict()
    for did, name, val in rows_of(d, counter):
        if "ocr::" in name or "_ZN3ocr" in name:
            per[did] = (pretty(name), per.get(did, ("", 0.0))[1] + val)
    return [per[k] for k in sorted(per)]


LABELS = None


def second_half(rows, labelled=True):
    rows = rows[len(rows) // 2:]           # two identical forwards per run: keep the second (warm) one
    if LABELS is not None and labelled:
        if len(rows) != len(LABELS):
            raise SystemExit(f"--labels: {len(LABELS)} table rows for {len(rows)} dispatches of a forward")
        rows = [(lab, val) for lab, (_, val) in zip(LABELS, rows)]
    return rows


def per_label(rows):
    agg = collections.OrderedDict()
    for name, val in rows:
        e = agg.setdefault(name, [0.0, 0])
        e[0] += val
        e[1] += 1
    return agg


def main():
    global LABELS
    sys.path.insert(0, ROOT)
    import bench  # (csrc_hash)
    args = sys.argv[1:]
    if "--labels" in args:
        i = args.index("--labels")
        LABELS = [m.group(1) for m in (re.match(r"\s*\d+\s+(\S+)\s+[0-9.]+\s", ln) for ln in open(args[i + 1])) if m]
        del args[i:i + 2]
    dtype, d_fetch, d_write, d_mops, d_busy = args[:5]
    tag = args[5] if len(args) > 5 else "r04"
    fetch = per_label(second_half(load(d_fetch, "FETCH_SIZE")))
    write = per_label(second_half(load(d_write, "WRITE_SIZE")))
    mf32 = per_label(second_half(load(d_mops, "SQ_INSTS_VALU_MFMA_MOPS_F32")))
    mbf16 = per_label(second_half(load(d_mops, "SQ_INSTS_VALU_MFMA_MOPS_BF16")))
    busy = per_label(second_half(load(d_busy, "SQ_VALU_MFMA_BUSY_CYCLES")))
    gui = per_label(second_half(load(d_busy, "GRBM_GUI_ACTIVE")))
    sha_file = os.path.join(os.path.dirname(os.path.abspath(d_fetch)), "csrc_sha.txt")   # written on the GPU box by collect_profiles.sh
    csrc_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else bench.csrc_hash()
    entry = {"batch": 32, "size": 640, "csrc_sha": csrc_sha,
             "source": f"rocprofv3 --pmc, one counter set per pass over tools/profile_layers.py 32 640 1 ({dtype}); profiles/{tag}_pmc_{dtype}.csv",
             "kernels": {}}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{dtype}.csv"), "w") as fcsv:
        fcsv.write("kernel,launches,fetch_bytes_per_launch_corrected,write_bytes_per_launch,mfma_flops_f32_per_launch,mfma_flops_bf16_per_launch,mfma_busy\n")
        for k, (f, n) in fetch.items():
            w = write.get(k, [0.0, n])[0]
            ff = mf32.get(k, [0.0, n])[0] * 512
            fb = mbf16.get(k, [0.0, n])[0] * 512
            bz = busy.get(k, [0.0, n])[0]
            g = gui.get(k, [0.0, n])[0]
            share = bz / (1024.0 * g / 8.0) if g > 0 else None
            entry["kernels"][k] = {"launches": n, "fetch_bytes": round(f * 1024 * 2 / n), "write_bytes": round(w * 1024 / n),
                                   "hbm_bytes": round((f * 1024 * 2 + w * 1024) / n), "mfma_flops": round((ff + fb) / n),
                                   "mfma_flops_f32": round(ff / n), "mfma_flops_bf16": round(fb / n),
                                   "mfma_busy": None if share is None else round(share, 4)}
            fcsv.write(f'"{k}",{n},{f * 1024 * 2 / n:.0f},{w * 1024 / n:.0f},{ff / n:.0f},{fb / n:.0f},{"" if share is None else f"{share:.4f}"}\n')
    if len(args) > 7:   # recogniser passes (tools/profile_rec.py 65536): fetch dir, write dir
        rf, rw = per_label(second_half(load(args[6], "FETCH_SIZE"), False)), per_label(second_half(load(args[7], "WRITE_SIZE"), False))
        entry["recogniser_b65536"] = {("rec_fc1" if k.startswith("conv_igemm") else k): {"launches": n, "hbm_bytes": round((f * 1024 * 2 + rw.get(k, [0.0, n])[0] * 1024) / n)}
                                      for k, (f, n) in rf.items()}
    path = os.path.join(ROOT, "profiles", "pmc.json")
    try:
        allp = json.load(open(path))
    except Exception:
        allp = {}
    allp[dtype] = entry
    json.dump(allp, open(path, "w"), indent=1)
    print(json.dumps(entry["kernels"], indent=1))


if __name__ == "__main__":
    main()

#!/bin/bash
# A/B of two builds of the library on ONE box: the layer table of each, alternating, three rounds.
#   tools/ab_libs.sh <libdir A> <libdir B> [engine options]   (directories under ocr-rs_amd/, e.g. lib_base lib)
A=$1; B=$2; OPT=$3; O=gpurun_out/ab; mkdir -p $O
for r in 1 2 3; do
  OCR_AMD_LIB=ocr-rs_amd/$A/libocr_amd.so timeout -k 10 200 python3 tools/profile_layers.py 32 640 5 0 "$OPT" > $O/a$r.txt 2>&1 || exit 1
  OCR_AMD_LIB=ocr-rs_amd/$B/libocr_amd.so timeout -k 10 200 python3 tools/profile_layers.py 32 640 5 0 "$OPT" > $O/b$r.txt 2>&1 || exit 1
done
python3 - $O <<'PY'
import sys, re, collections
o = sys.argv[1]
def load(f):
    rows = []
    for l in open(f):
        m = re.match(r"\s*(\d+) (\S+)\s+([\d.]+)", l)
        if m: rows.append((int(m.group(1)), m.group(2), float(m.group(3))))
    return rows
a = [load(f"{o}/a{r}.txt") for r in (1, 2, 3)]
b = [load(f"{o}/b{r}.txt") for r in (1, 2, 3)]
ta = tb = 0.0
for i in range(len(a[0])):
    ma = min(x[i][2] for x in a); mb = min(x[i][2] for x in b)
    ta += ma; tb += mb
    if "wino" not in a[0][i][1] or "fused" in a[0][i][1]:
        print(f"{i:2d} {a[0][i][1]:46s} {ma:.4f} {mb:.4f} {100 * (mb / ma - 1):+.1f}%")
print(f"total (min of 3 per row) {ta:.3f} {tb:.3f} {100 * (tb / ta - 1):+.2f}%")
PY

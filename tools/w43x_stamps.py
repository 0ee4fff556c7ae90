#!/usr/bin/env python3
"""Diagnostic (library built with make EXTRA=-DW43_STAMPS): phase timeline of workgroup 0 of one winograd43_x3 launch
(64 -> 64 at 160 x 160, batch 32), waves 0 (half 0) and 4 (half 1), first three units, in shader cycles (s_memtime)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = 160 if cin == 64 else 80
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
rng = np.random.default_rng(0)
x = np.maximum(rng.standard_normal((32, hw, hw, cin), dtype=np.float32), 0)
wg = (rng.standard_normal((cin, 9, cin), dtype=np.float32) / 24).astype(np.float32)
res = rng.standard_normal((32, hw, hw, cin), dtype=np.float32)
for _ in range(2):
    det.debug_winograd_conv(x, wg, None, None, res, True, unfused=5)
out = (ctypes.c_longlong * (2 * 3 * 64))()
capi.test_lib().ocr_test_w43x_stamps(out)
a = np.array(out[:]).reshape(2, 3, 64)
names = {0: "unit start"}
for c in range(4):
    b = 14 * c
    names[b + 1] = f"c{c} ring prologue issued"
    names[b + 2] = f"c{c} patch landed (own share)"
    names[b + 3] = f"c{c} top barrier passed"
    for ph in range(3):
        names[b + 4 + 4 * ph] = f"c{c} T{ph} done"
        names[b + 5 + 4 * ph] = f"c{c} T{ph} barrier passed"
        names[b + 6 + 4 * ph] = f"c{c} M{ph} done"
        if ph < 2:
            names[b + 7 + 4 * ph] = f"c{c} M{ph} barrier passed"
names[57] = "epilogue start"
names[58] = "exchange done (Y in registers)"
names[59] = "h0 residual requested"
names[61] = "h1 residual requested"
names[63] = "unit end"
for wv in range(2):
    for u in range(3):
        t0 = a[wv, u, 0]
        if t0 == 0:
            continue
        print(f"--- wave {4 * wv} unit {u} (start {t0 - a[0, 0, 0]})")
        prev = t0
        for k in sorted(names):
            t = a[wv, u, k]
            if t == 0:
                continue
            print(f"  {names[k]:36s} +{t - t0:7d}  (d {t - prev:6d})")
            prev = t

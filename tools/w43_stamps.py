#!/usr/bin/env python3
"""Diagnostic (library built with make EXTRA=-DW43_STAMPS): phase timeline of workgroup 0 of one fused F(4x4,3x3) launch
(64 -> 64 at 160 x 160, batch 32), in s_memtime ticks (100 MHz: 10 ns) relative to the block's start."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library  # noqa: E402

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
rng = np.random.default_rng(0)
x = np.maximum(rng.standard_normal((32, 160, 160, 64), dtype=np.float32), 0)
wg = (rng.standard_normal((64, 9, 64), dtype=np.float32) / 24).astype(np.float32)
res = rng.standard_normal((32, 160, 160, 64), dtype=np.float32)
for _ in range(2):
    det.debug_winograd_conv(x, wg, None, None, res, True, unfused=4)
out = (ctypes.c_longlong * (2 * 4 * 32 + 2 * 4 * 26))()
capi.test_lib().ocr_test_w43_stamps(out)
a = np.array(out[:2 * 4 * 32]).reshape(2, 4, 32)
steps = np.array(out[2 * 4 * 32:]).reshape(2, 4, 26)
names = {0: "block start"}
for c in range(4):
    names[1 + 4 * c] = f"c{c} before top barrier"
    names[2 + 4 * c] = f"c{c} after top barrier"
    names[3 + 4 * c] = f"c{c} transform done"
    names[4 + 4 * c] = f"c{c} V barrier passed"
for h in range(2):
    names[17 + 4 * h] = f"h{h} mfma done / residual requested"
    names[18 + 4 * h] = f"h{h} barrier passed"
    names[19 + 4 * h] = f"h{h} out transform + staging done"
    names[20 + 4 * h] = f"h{h} staging barrier passed"
names[25] = "block end"
for wv in range(2):
    for b in range(4):
        t0 = a[wv, b, 0]
        print(f"--- wave {wv} block {b} (start tick {t0 - a[0, 0, 0]})")
        prev = t0
        for k in sorted(names):
            t = a[wv, b, k]
            if t == 0:
                continue
            print(f"  {names[k]:40s} +{t - t0:6d}  (d {t - prev:5d})")
            prev = t

print("--- matrix phases of block 2: per step, cycles waiting for B | cycles from the wait to the next step's wait")
for wv in range(2):
    for c in range(4):
        st = steps[wv, c]
        if st[0] == 0:
            continue
        waits = [int(st[2 * t + 1] - st[2 * t]) for t in range(12)]
        body = [int(st[2 * (t + 1)] - st[2 * t + 1]) for t in range(11)]
        print(f"  wave {wv} chunk {c}: wait {waits}  body {body}")

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
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64      # 64: the H/4 grid (160 x 160), 128: the H/8 grid (80 x 80)
HW = 160 if C == 64 else 80
NCH = C // 16
x = np.maximum(rng.standard_normal((32, HW, HW, C), dtype=np.float32), 0)
wg = (rng.standard_normal((C, 9, C), dtype=np.float32) / 24).astype(np.float32)
res = rng.standard_normal((32, HW, HW, C), dtype=np.float32)
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

# ---- summary: where a block's cycles go (wave 0, blocks 1..3; chunks beyond the fourth are not stamped: c128 shows its first four of eight)
print("--- summary per block (ticks of s_memtime; wave 0, mean of blocks 1-3)")
tot = pre = tr = bar = mat = epi = 0.0
nb = 0
for b in range(1, 4):
    t = a[0, b]
    if t[25] == 0:
        continue
    nb += 1
    tot += t[25] - t[0]
    pre += t[1] - t[0]
    for c in range(4):
        bar += (t[2 + 4 * c] - t[1 + 4 * c]) + (t[4 + 4 * c] - t[3 + 4 * c])
        tr += t[3 + 4 * c] - t[2 + 4 * c]
        nxt = t[5 + 4 * c] if c < 3 else t[17]
        if NCH == 4 or c < 3:
            mat += nxt - t[4 + 4 * c]
    epi += t[25] - t[17]
if nb:
    st = steps[0]
    waits = np.mean([[st[c][2 * k + 1] - st[c][2 * k] for k in range(12)] for c in range(4) if st[c][0]])
    body = np.mean([[st[c][2 * (k + 1)] - st[c][2 * k + 1] for k in range(11)] for c in range(4) if st[c][0]])
    print(f"block {tot / nb:.0f} ticks: before chunk 0 {pre / nb:.0f}, transforms (4 chunks) {tr / nb:.0f}, barriers (8) {bar / nb:.0f}, "
          f"matrix phases (stamped chunks) {mat / nb:.0f}, epilogue {epi / nb:.0f}; per matrix step: B wait {waits:.0f}, body {body:.0f} (12 MFMA = 384 cycles of matrix pipe)")

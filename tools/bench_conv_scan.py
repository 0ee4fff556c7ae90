#!/usr/bin/env python3
"""Scan of conv_igemm time vs K (Cin) and vs M (batch) to separate per-launch, per-tile and per-K-step cost."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
print("3x3 s1, 160x160, Cout=64 (tile 128x64)")
for n in (4, 8, 16, 32):
    row = []
    for ci in (64, 128, 256, 512):
        if n * 160 * 160 * ci * 4 >= 2**31:
            row.append("   -   ")
            continue
        ms = det.debug_conv_bench(n, 160, 160, ci, 64, 3, 1, 0, 10)
        row.append(f"{ms*1e3:7.1f}")
    print(f"N={n:2d} us at Cin=64/128/256/512: " + " ".join(row))
print("3x3 s1, 40x40, Cout=256 (tile 64x64)")
for n in (8, 16, 32, 64):
    row = []
    for ci in (64, 128, 256, 512):
        ms = det.debug_conv_bench(n, 40, 40, ci, 256, 3, 1, 0, 10)
        row.append(f"{ms*1e3:7.1f}")
    print(f"N={n:2d} us at Cin=64/128/256/512: " + " ".join(row))

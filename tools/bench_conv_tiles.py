#!/usr/bin/env python3
"""Tile-shape scan of conv_igemm on the detector's layer shapes: python tools/bench_conv_tiles.py [bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library  # noqa: E402

bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
shapes = [
    ("layer1 conv 64->64", 32, 160, 160, 64, 64, 3, 1),
    ("layer2 conv 128->128", 32, 80, 80, 128, 128, 3, 1),
    ("layer2.0 conv1 s2", 32, 160, 160, 64, 128, 3, 2),
    ("layer3 conv 256->256", 32, 40, 40, 256, 256, 3, 1),
    ("layer3.0 conv1 s2", 32, 80, 80, 128, 256, 3, 2),
    ("layer4 conv 512->512", 32, 20, 20, 512, 512, 3, 1),
    ("layer4.0 conv1 s2", 32, 40, 40, 256, 512, 3, 2),
    ("out4 256->64 @40", 32, 40, 40, 256, 64, 3, 1),
    ("p3A 128->64 @80", 32, 80, 80, 128, 64, 3, 1),
]
print(f"{'shape':26s} " + " ".join(f"{t:>16s}" for t in ("128x128", "128x64", "64x64")))
for lab, n, h, w, ci, co, ks, st in shapes:
    ho, wo = (h + 2 * (ks // 2) - ks) // st + 1, (w + 2 * (ks // 2) - ks) // st + 1
    fl = 2.0 * n * ho * wo * co * ks * ks * ci
    cells = []
    for t in (1, 2, 3):
        if t == 1 and co % 128:
            cells.append(f"{'-':>16s}")
            continue
        capi.test_lib().ocr_test_set_conv_tile(t)
        ms = det.debug_conv_bench(n, h, w, ci, co, ks, st, 16 if bf else 0, 10)
        cells.append(f"{ms:7.4f}ms {fl / ms / 1e9:6.1f}")
    print(f"{lab:26s} " + " ".join(cells))
capi.test_lib().ocr_test_set_conv_tile(0)

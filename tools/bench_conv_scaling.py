#!/usr/bin/env python3
"""conv_igemm rate against launch size (ramp/drain share) on trunk shapes: python tools/bench_conv_scaling.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
mode = 32 if len(sys.argv) > 1 and sys.argv[1] == "random" else 0   # operands: constants or random values
for lab, h, ci, co in (("layer1", 160, 64, 64), ("layer2", 80, 128, 128), ("layer3", 40, 256, 256), ("layer4", 20, 512, 512)):
    row = []
    for n in (8, 32, 64):
        ms = det.debug_conv_bench(n, h, h, ci, co, 3, 1, mode, 10)
        fl = 2.0 * n * h * h * co * 9 * ci
        row.append(f"n={n}: {ms:7.4f}ms {fl / ms / 1e9:6.1f}")
    print(f"{lab:8s} " + " | ".join(row))

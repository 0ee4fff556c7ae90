#!/usr/bin/env python3
"""Micro-benchmark of single conv_igemm shapes: python tools/bench_conv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
shapes = [
    # label, n, h, w, cin, cout, ks, stride
    ("steady 512 blk 128x128 K=4608", 16, 64, 64, 512, 128, 3, 1),
    ("steady 1024 blk 128x128 K=4608", 32, 64, 64, 512, 128, 3, 1),
    ("steady 512 blk 128x64  K=4608", 8, 64, 64, 512, 64, 3, 1),
    ("steady 1024 blk 128x64 K=4608", 16, 64, 64, 512, 64, 3, 1),
    ("layer1 conv", 32, 160, 160, 64, 64, 3, 1),
    ("layer2 conv", 32, 80, 80, 128, 128, 3, 1),
    ("layer3 conv", 32, 40, 40, 256, 256, 3, 1),
    ("layer4 conv", 32, 20, 20, 512, 512, 3, 1),
    ("out2-like plain 256->64", 32, 160, 160, 256, 64, 3, 1),
]
for lab, n, h, w, ci, co, ks, st in shapes:
    ms = det.debug_conv_bench(n, h, w, ci, co, ks, st, 0, 5)
    ho, wo = (h + 2 * (ks // 2) - ks) // st + 1, (w + 2 * (ks // 2) - ks) // st + 1
    fl = 2.0 * n * ho * wo * co * ks * ks * ci
    print(f"{lab:34s} {ms:8.4f} ms  {fl / ms / 1e9:7.1f} TF/s")

#!/usr/bin/env python3
"""Diagnostic (library built with make EXTRA=-DSTEM_STAMPS): phase timeline of one workgroup of the split-bf16 stem (batch 32, 640 x 640),
waves 0 and 3, its four tiles, in shader cycles."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
for _ in range(3):
    det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr())
torch.cuda.synchronize()
out = (ctypes.c_longlong * (2 * 4 * 8))()
capi.test_lib().ocr_test_stem_stamps(out)
a = np.array(out[:]).reshape(2, 4, 8)
names = ["tile start", "staged (split + LDS writes issued, next tile requested)", "barrier 1 passed", "MFMAs issued", "conv tile written to LDS",
         "barrier 2 passed", "pooled + stored"]
for wv in range(2):
    t00 = a[wv, 0, 0]
    for tl in range(4):
        print(f"--- wave {3 * wv} tile {tl} (start +{a[wv, tl, 0] - t00})")
        prev = a[wv, tl, 0]
        for k in range(1, 7):
            print(f"  {names[k]:58s} +{a[wv, tl, k] - a[wv, tl, 0]:6d} (d {a[wv, tl, k] - prev:5d})")
            prev = a[wv, tl, k]

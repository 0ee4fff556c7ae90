#!/usr/bin/env python3
"""Per-step timeline of winograd_ws_kernel (workgroup 0): needs a diagnostic build,
    make -C ocr-rs_amd/csrc EXTRA=-DWS_STAMPS OUT=../lib_stamps && OCR_AMD_LIB=ocr-rs_amd/lib_stamps/libocr_amd.so python tools/ws_stamps.py [cin]
Prints, per step: multiplier busy (barrier exit -> next barrier arrival), helper busy, and the step period."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library  # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
rng = np.random.default_rng(0)
n, h, w, cout = 32, {64: 160, 128: 80, 256: 40}[cin], {64: 160, 128: 80, 256: 40}[cin], cin
x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
res = rng.standard_normal((n, h, w, cout), dtype=np.float32)
for _ in range(2):
    det.debug_winograd_conv(x, wg, np.ones(cout, np.float32), np.zeros(cout, np.float32), res, True, unfused=2)
buf = (C.c_longlong * (2 * 128 * 4))()
capi.test_lib().ocr_test_ws_stamps(buf)
a = np.array(buf[:], dtype=np.int64).reshape(2, 128, 4)
print("step  mult_busy  help_busy  period(mult) | helper: transform  epilogue  dma+loads+waits")
for s in range(1, 40):
    print(f"{s:4d} {a[0, s, 1] - a[0, s, 0]:10d} {a[1, s, 1] - a[1, s, 0]:10d} {a[0, s, 0] - a[0, s - 1, 0]:10d} | "
          f"{a[1, s, 2] - a[1, s, 0]:10d} {a[1, s, 3] - a[1, s, 2]:10d} {a[1, s, 1] - a[1, s, 3]:10d}")

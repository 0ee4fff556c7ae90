#!/usr/bin/env python3
"""Diagnostic (library built with make EXTRA=-DREC_STAMPS): s_memtime deltas of workgroup 0 of rec_conv_small_kernel at 256 crops:
load | conv1 | barrier | conv2 | store.  Measured: 1600, 3956, 2612, 23336, 640 cycles - conv2 is 16 waves x 200 MFMAs of 32
cycles on 4 SIMDs = 25.6 k cycles: one crop per CU is bound by the CU's matrix rate (10.7 us at the peak)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library
rec = capi.Recognizer(W.pack_blob(W.make_rec_weights(0)), 0)
nc = 256
crops = torch.from_numpy(W.synth_crops(2, nc)).cuda()
labels = torch.empty(nc, dtype=torch.int32, device="cuda"); probs = torch.empty(nc, dtype=torch.float64, device="cuda")
for _ in range(20):
    rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
torch.cuda.synchronize()
out = (ctypes.c_longlong * 16)()
capi.test_lib().ocr_test_rec_stamps(out)
a = list(out)
print("deltas:", [a[i + 1] - a[i] for i in range(5)], "total", a[5] - a[0])

#!/usr/bin/env python3
"""Tuning aid (library built with make EXTRA=-DW43_DEBUG): per-launch time of the fused F(4x4,3x3) kernel with parts of it switched off (ocr_test_w43_debug bits:
1 no B loads, 2 no input transform, 4 no patch DMA, 8 no stores).  usage: python tools/w43_probe.py [options]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library  # noqa: E402

opts = sys.argv[1] if len(sys.argv) > 1 else None
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options=opts)
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
for dbg in (0, 1, 2, 4, 8, 3, 7, 15):
    capi.test_lib().ocr_test_w43_debug(dbg)
    acc = {}
    for r in range(4):
        prof = det.forward_profile(x.data_ptr(), 32, 640, 640, prob.data_ptr())
        if r == 0:
            continue
        for i, (nm, ms, fl, by) in enumerate(prof):
            if nm.startswith("winograd43_fused"):
                acc.setdefault((i, nm), []).append(ms)
    print(f"debug={dbg:2d} " + "  ".join(f"{nm[17:]}#{i}:{sum(v)/len(v):.4f}" for (i, nm), v in sorted(acc.items())[:8]))
capi.test_lib().ocr_test_w43_debug(0)

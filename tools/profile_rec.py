#!/usr/bin/env python3
"""Two classify passes of the recogniser over N device-resident crops (for rocprofv3 --pmc / --kernel-trace):
python tools/profile_rec.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rec = capi.Recognizer(W.pack_blob(W.make_rec_weights(0)), 0)
crops = torch.from_numpy(W.synth_crops(2, nc)).cuda()
labels = torch.empty(nc, dtype=torch.int32, device="cuda")
probs = torch.empty(nc, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
rec.synchronize()
print("labels", labels[:8].tolist())

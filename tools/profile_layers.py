#!/usr/bin/env python3
"""Per-launch timing of one detector forward (HIP events around every launch).
usage: python tools/profile_layers.py [batch] [size] [reps] [tile override or 0] [engine options, e.g. precision=bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = int(sys.argv[2]) if len(sys.argv) > 2 else 640
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options=sys.argv[5] if len(sys.argv) > 5 else None)
if len(sys.argv) > 4 and int(sys.argv[4]):
    capi.test_lib().ocr_test_set_conv_tile(int(sys.argv[4]))   # tuning aid: force a conv_igemm tile (1, 2, 3)
x = torch.from_numpy(W.synth_image_batch(1, n, s, s)).cuda()
prob = torch.empty_like(x)
torch.cuda.synchronize()
acc = None
for r in range(reps + 1):
    prof = det.forward_profile(x.data_ptr(), n, s, s, prob.data_ptr())
    if r == 0:
        continue  # warm-up
    if acc is None:
        acc = [[nm, 0.0, fl, by] for nm, ms, fl, by in prof]
    for i, (nm, ms, fl, by) in enumerate(prof):
        acc[i][1] += ms
tot = 0.0
print(f"{'#':>2} {'kernel':<46} {'ms':>8} {'GFLOP':>9} {'TF/s':>7} {'GB/s':>8}")
for i, (nm, ms, fl, by) in enumerate(acc):
    ms /= reps
    tot += ms
    print(f"{i:>2} {nm:<46} {ms:8.4f} {fl / 1e9:9.2f} {fl / ms / 1e9:7.1f} {by / ms / 1e6:8.1f}")
print(f"total {tot:.3f} ms  -> {n / tot * 1e3:.1f} img/s, {sum(a[2] for a in acc) / tot / 1e9:.1f} TF/s")

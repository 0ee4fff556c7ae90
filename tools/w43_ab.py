"""A/B of the fused F(4x4,3x3) kernel's block order in ONE process (devices differ by several per cent): XCD-contiguous
runs against the plain linear order, alternating forwards.  python3 tools/w43_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
acc = {0: {}, 16: {}}
for r in range(rounds + 1):
    for dbg in (0, 16):
        capi.test_lib().ocr_test_w43_debug(dbg)
        for nm, ms, fl, by in det.forward_profile(x.data_ptr(), 32, 640, 640, prob.data_ptr()):
            if r and "winograd43_fused" in nm:
                e = acc[dbg].setdefault(nm, [0.0, 0])
                e[0] += ms
                e[1] += 1
capi.test_lib().ocr_test_w43_debug(0)
for nm in acc[0]:
    a, b = acc[0][nm], acc[16][nm]
    print(f"{nm:28s} xcd runs {a[0] / a[1]:.4f} ms   linear {b[0] / b[1]:.4f} ms   ({a[1]} launches each)")

#!/usr/bin/env python3
"""Back-to-back forward time under the second-stream options (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
blob = W.pack_blob(W.make_det_weights(0))
opts = sys.argv[1:] or [None, "overlap=2", "overlap=2;w43_side_cus=128", "overlap=3", "overlap=3;w43_side_cus=192", "overlap=3;w43_side_cus=128", "overlap=3;w43_side_cus=96", "overlap=3;w43_side_cus=64", None]
for opt in opts:
    opt = None if opt in (None, "none") else opt
    det = capi.Detector(blob, 0, options=opt)
    for _ in range(5): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    torch.cuda.synchronize(); det.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(20): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
        det.synchronize(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 20 * 1e3)
    print(f"{str(opt):40s} {best:.3f} ms", flush=True)
    det.close()

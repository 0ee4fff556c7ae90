#!/usr/bin/env python3
"""Back-to-back forward time under the second-stream options (run on the GPU box): default 5.92 ms, overlap=1 5.97, overlap=2 5.96 -
the small independent launches gain nothing next to the large ones on this schedule; the options stay opt-in."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
blob = W.pack_blob(W.make_det_weights(0))
for opt in (None, "overlap=1", "overlap=2"):
    det = capi.Detector(blob, 0, options=opt)
    for _ in range(5): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    torch.cuda.synchronize(); det.synchronize()
    t = time.perf_counter()
    for _ in range(30): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    det.synchronize(); torch.cuda.synchronize()
    print(opt, round((time.perf_counter() - t) / 30 * 1e3, 3), "ms")
    det.close()

#!/usr/bin/env python3
"""bf16 precision: the step with and without the side-stream schedule (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
blob = W.pack_blob(W.make_det_weights(0))
for opt in ("precision=bf16", "precision=bf16;overlap=0", "precision=bf16", "precision=bf16;overlap=0"):
    det = capi.Detector(blob, 0, options=opt)
    for _ in range(5): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    torch.cuda.synchronize(); det.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(30): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
        det.synchronize(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 30 * 1e3)
    print(f"{str(opt):40s} {best:.3f} ms", flush=True)
    det.close()

"""Step time of the detector (32 x 640 x 640, device-resident) under engine options, alternating in ONE process:
python3 tools/bench_options.py "opt1" "opt2" ...   ("" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
opts = sys.argv[1:] or ["", "overlap=1", "overlap=2"]
blob = W.pack_blob(W.make_det_weights(0))
dets = [capi.Detector(blob, 0, options=o or None) for o in opts]
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
acc = [0.0] * len(opts)
R = 5
for r in range(R + 1):
    for i, d in enumerate(dets):
        for _ in range(3):
            d.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr())
        d.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr())
        d.synchronize()
        if r:
            acc[i] += (time.perf_counter() - t0) / 10
for o, a in zip(opts, acc):
    print(f"{o or 'default':40s} {a / R * 1e3:.3f} ms per step")

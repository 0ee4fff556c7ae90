#!/usr/bin/env python3
"""BASELINE configs[4] (128 pages, bf16 trunk with f32 accumulate) against the f32 path on the same frames:
probability-map distance, binarisation flips at the 0.6 threshold, frame rate of both precisions.
Polygon / label mismatch counts need trained weights (random-weight maps are noise without text), so the
post-network stages are exercised on the f32 maps elsewhere (tests/test_gpu_pipeline.py).
usage: python tools/cfg5_report.py [pages] [size] > profiles/r01_cfg5_bf16.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

pages = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = int(sys.argv[2]) if len(sys.argv) > 2 else 640
bs = 32
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
stats = {"pages": pages, "size": s, "max_abs": 0.0, "sum_abs": 0.0, "flips": 0, "pixels": 0}
rate = {}
for b in range(0, pages, bs):
    n = min(bs, pages - b)
    x = torch.from_numpy(W.synth_image_batch(100 + b, n, s, s)).cuda()
    out = {}
    for name, prec in (("f32", capi.PRECISION_F32), ("bf16", capi.PRECISION_BF16)):
        det.set_precision(prec)
        prob = torch.empty_like(x)
        det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr())
        det.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr())
        det.synchronize()
        rate.setdefault(name, []).append(n * 5 / (time.perf_counter() - t0))
        out[name] = prob
    d = (out["bf16"] - out["f32"]).abs()
    stats["max_abs"] = max(stats["max_abs"], float(d.max()))
    stats["sum_abs"] += float(d.double().sum())
    stats["flips"] += int(((out["bf16"] > 0.6) != (out["f32"] > 0.6)).sum())
    stats["pixels"] += d.numel()
print(json.dumps({"config": "BASELINE configs[4]: bf16 trunk/FPN, f32 accumulate, f32 head",
                  "pages": pages, "frame": [s, s], "weights": "synthetic seed 0",
                  "max_abs_dprob": stats["max_abs"], "mean_abs_dprob": stats["sum_abs"] / stats["pixels"],
                  "binarize_flips_at_0.6": stats["flips"], "pixels": stats["pixels"],
                  "flip_fraction": stats["flips"] / stats["pixels"],
                  "frames_per_s": {k: round(sum(v) / len(v), 1) for k, v in rate.items()}}))

#!/usr/bin/env python3
"""Recognition throughput vs batch, with the per-kernel split: python tools/bench_rec.py [sizes...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [1, 64, 256, 1024, 4096, 16384, 65536]
rec = capi.Recognizer(W.pack_blob(W.make_rec_weights(0)), 0)
for nc in sizes:
    crops = torch.from_numpy(W.synth_crops(2, nc)).cuda()
    labels = torch.empty(nc, dtype=torch.int32, device="cuda")
    probs = torch.empty(nc, dtype=torch.float64, device="cuda")
    for _ in range(3):
        rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
    rec.synchronize()
    it = 30 if nc <= 16384 else 5
    t0 = time.perf_counter()
    for _ in range(it):
        rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
    rec.synchronize()
    dt = (time.perf_counter() - t0) / it
    prof = rec.classify_profile(crops.data_ptr(), nc, labels.data_ptr(), probs.data_ptr())
    split = "  ".join(f"{n}={ms * 1e3:.1f}us" + (f"({fl / ms / 1e9:.0f}TF)" if fl else "") for n, ms, fl, _ in prof)
    print(f"N={nc:6d}: {dt * 1e6:9.1f} us  {nc / dt / 1e6:7.2f} M crops/s  {nc * 8.587264 / dt / 1e6:6.1f} TFLOP/s | {split}", flush=True)

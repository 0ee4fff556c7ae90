#!/usr/bin/env python3
"""Post-processing throughput on text-like and dense maps: python tools/postproc_time.py [post_threads, default 0 = automatic] [more engine options]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
from tests import fixtures as FX
pt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
extra = sys.argv[2] if len(sys.argv) > 2 else ""
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options=f"post_threads={pt};{extra}")
n, s = 32, 640
params = capi.default_params(skip_degenerate=True)
adj = np.ones((n, 2))
for name, maps in (("text-like", FX.text_like_maps(n, s, 0)), ("dense", FX.dense_text_maps(n, s, 5))):
    pm = torch.from_numpy(maps).cuda(); torch.cuda.synchronize()
    polys, _ = det.postprocess(pm, n, s, s, adj, capi.MEM_DEVICE, params)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter()
        det.postprocess_counts(pm, n, s, s, adj, capi.MEM_DEVICE, params)
        best = min(best, time.perf_counter() - t0)
    print(f"{name}: {sum(len(p) for p in polys) / n:.1f} polygons/image, post_threads={pt} {extra}: {best * 1e3:.2f} ms per {n} maps = {n / best:.0f} images/s", flush=True)

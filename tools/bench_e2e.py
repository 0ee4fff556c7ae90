#!/usr/bin/env python3
"""End-to-end detection throughput: forward (+fused binarize) then get_boxes_and_box_scores on
realistic probability maps.  Random-weight maps are noise (thousands of specks per frame), so the
post-processing leg is fed tiled copies of the reference's gt_shrinked fixtures:
text-like blobs, 640x640, values jittered around 0.9 / 0.05.
usage: python tools/bench_e2e.py [batch] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = 640
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
rng = np.random.RandomState(0)
maps = []
names = ["gt_shrinked_img55.png", "gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]
for i in range(n):
    g = np.array(Image.open(os.path.join(ROOT, "tests", "golden", names[i % 4])).convert("L"))[80:720, 80:720]
    if (i // 4) % 2:
        g = g[:, ::-1]
    m = np.where(g > 127, 0.8 + 0.2 * rng.rand(s, s), 0.1 * rng.rand(s, s)).astype(np.float32)
    maps.append(m)
pred_host = np.ascontiguousarray(np.stack(maps)[:, None])
pred_dev = torch.from_numpy(pred_host).cuda()
adj = np.ones((n, 2))
x = torch.from_numpy(W.synth_image_batch(1, n, s, s)).cuda()
prob = torch.empty_like(x)
torch.cuda.synchronize()
params = capi.default_params(skip_degenerate=True)

# warm up
det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr()); det.synchronize()
polys, scores = det.postprocess(pred_dev, n, s, s, adj, capi.MEM_DEVICE, params)
npoly = sum(len(p) for p in polys)

t0 = time.perf_counter()
for _ in range(iters):
    det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr())
det.synchronize()
t_fwd = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
for _ in range(iters):
    det.postprocess(pred_dev, n, s, s, adj, capi.MEM_DEVICE, params)
t_post = (time.perf_counter() - t0) / iters
t0 = time.perf_counter()
for _ in range(iters):
    det.postprocess(pred_host, n, s, s, adj, capi.MEM_HOST, params)
t_post_h = (time.perf_counter() - t0) / iters
print(f"batch {n}: forward {t_fwd * 1e3:.2f} ms ({n / t_fwd:.0f} img/s) | postprocess(device map) {t_post * 1e3:.2f} ms "
      f"({n / t_post:.0f} img/s, {npoly} polygons) | postprocess(host map, incl. H2D) {t_post_h * 1e3:.2f} ms | "
      f"serial forward+post {n / (t_fwd + t_post):.0f} img/s")

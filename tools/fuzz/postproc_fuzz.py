import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ocr_rs_amd
from ocr_rs_amd import weights as W
from oracle import postproc_cpu as PC
rng = np.random.default_rng(5)
tot = 0
# text-like and dense pages
for dense in (False, True):
    pages = W.synth_text_pages(3, 4, 640, 640, dense=dense)[0]
    m = pages.astype(np.float32)
    m = (m - m.min()) / (m.max() - m.min() + 1e-9)
    p, s = PC.get_boxes_and_box_scores(m, np.ones((4, 2)), threads=4)
    tot += sum(len(x) for x in p)
# noise at several densities, smoothed noise, stripes, checkerboards, full and empty maps, thin lines, touching the borders
for it in range(60):
    h, w = int(rng.choice([32, 64, 96, 160, 320])), int(rng.choice([32, 64, 128, 224, 320]))
    kind = it % 6
    if kind == 0: m = rng.random((2, 1, h, w), dtype=np.float32)
    elif kind == 1:
        m = rng.random((2, 1, h, w), dtype=np.float32)
        for _ in range(3): m = (m + np.roll(m, 1, 2) + np.roll(m, 1, 3) + np.roll(m, -1, 2) + np.roll(m, -1, 3)) / 5
        m = (m - m.min()) / (m.max() - m.min())
    elif kind == 2: m = np.ones((2, 1, h, w), np.float32)
    elif kind == 3: m = np.zeros((2, 1, h, w), np.float32); m[:, :, ::3] = 1; m[:, :, :, ::5] = 1
    elif kind == 4: m = ((np.add.outer(np.arange(h), np.arange(w)) // (1 + it % 4)) % 2).astype(np.float32)[None, None].repeat(2, 0)
    else:
        m = np.zeros((2, 1, h, w), np.float32)
        for _ in range(12):
            y0, x0 = rng.integers(0, h), rng.integers(0, w); hh, ww = rng.integers(1, 40), rng.integers(1, 80)
            m[:, :, y0:y0 + hh, x0:x0 + ww] = rng.random() * 0.5 + 0.5
    adj = rng.random((2, 2)) * 3 + 0.3
    for thr in (1, 3):
        p, s = PC.get_boxes_and_box_scores(m, adj, threads=thr, skip_degenerate=True, min_size=float(rng.choice([0.0, 3.0, 5.0])), unclip_ratio=float(rng.choice([0.5, 1.5, 2.0, 4.0])))
        tot += sum(len(x) for x in p)
print("polygons:", tot)

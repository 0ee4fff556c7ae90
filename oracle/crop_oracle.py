"""ORACLE (test infrastructure) for the BUILD-DEFINED crop rule that links detection to recognition.

The reference never implemented this step ("Character Segmentation" is unchecked in README.md:20-26 and
pipeline.png); there is NO reference behaviour to match, so parity here is only "HIP kernel == this
restatement of our own rule".  Rule, for every kept polygon of frame b:
  * map its vertices back to frame coordinates: x_f = x * adj_x, y_f = y * adj_y (f64), take the axis-aligned
    bounding box [x0, x1] x [y0, y1] clamped to the frame, widened to at least one pixel;
  * sample a 28 x 28 grid with bilinear interpolation at pixel centres (half-pixel convention, edge clamp):
    sx = x0 + (j + 0.5) * (x1 - x0) / 28 - 0.5, likewise sy, f32 arithmetic;
  * divide by 255 (load_image_as_tensor's scaling, image_ops.rs:80-83): crops are N x 784 f32 in [0, 1].
"""
from __future__ import annotations

import numpy as np

F = np.float32


def crop_boxes(polys, adj, h, w):
    """Per polygon: (frame index, x0, y0, x1, y1) as f32, polygons in batch order."""
    out = []
    for b, plist in enumerate(polys):
        ax, ay = float(adj[b][0]), float(adj[b][1])
        for poly in plist:
            xs = [p[0] * ax for p in poly]
            ys = [p[1] * ay for p in poly]
            x0 = min(max(min(xs), 0.0), w - 1.0)
            x1 = min(max(max(xs) + 1.0, x0 + 1.0), float(w))
            y0 = min(max(min(ys), 0.0), h - 1.0)
            y1 = min(max(max(ys) + 1.0, y0 + 1.0), float(h))
            out.append((b, F(x0), F(y0), F(x1), F(y1)))
    return out


def extract_crops(frames: np.ndarray, polys, adj) -> np.ndarray:
    """frames: N x 1 x H x W f32 (raw 0..255).  Returns P x 784 f32."""
    n, _, h, w = frames.shape
    boxes = crop_boxes(polys, adj, h, w)
    out = np.zeros((len(boxes), 784), np.float32)
    for k, (b, x0, y0, x1, y1) in enumerate(boxes):
        img = frames[b, 0]
        sxs = F(x1 - x0) / F(28)
        sys_ = F(y1 - y0) / F(28)
        for i in range(28):
            sy = F(F(y0 + F(F(i) + F(0.5)) * sys_) - F(0.5))
            sy = min(max(sy, F(0)), F(h - 1))
            iy0 = int(np.floor(sy)); iy1 = min(iy0 + 1, h - 1); fy = F(sy - F(iy0))
            for j in range(28):
                sx = F(F(x0 + F(F(j) + F(0.5)) * sxs) - F(0.5))
                sx = min(max(sx, F(0)), F(w - 1))
                ix0 = int(np.floor(sx)); ix1 = min(ix0 + 1, w - 1); fx = F(sx - F(ix0))
                top = F(img[iy0, ix0] + F(fx * F(img[iy0, ix1] - img[iy0, ix0])))
                bot = F(img[iy1, ix0] + F(fx * F(img[iy1, ix1] - img[iy1, ix0])))
                out[k, i * 28 + j] = F(F(top + F(fy * F(bot - top))) / F(255))
    return out

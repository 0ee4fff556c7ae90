"""Map fixtures shared by the GPU tests, bench.py and the timing tools (NOT part of the product package: they read the
reference-held PNG fixtures under tests/golden).

text_like_maps  - probability maps cut from the reference's gt_shrinked_* label maps (/root/reference/test_data, copied
                  byte for byte into tests/golden)
dense_text_maps - synthetic post-processing stress pages
"""
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def text_like_maps(n: int, s: int, seed: int):
    """Probability maps with text-like blobs (the reference's gt_shrinked fixtures, cropped to s x s and
    jittered): random-weight network outputs are noise, which is not what post-processing sees in use."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    names = ["gt_shrinked_img55.png", "gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]
    base = [np.array(Image.open(os.path.join(_GOLDEN, nm)).convert("L")) for nm in names]
    out = []
    for i in range(n):
        g = base[i % 4]
        o = (800 - s) // 2
        g = g[o:o + s, o:o + s] if s <= 800 else np.pad(g, ((0, s - 800), (0, s - 800)))
        out.append(np.where(g > 127, 0.8 + 0.2 * rng.rand(s, s), 0.1 * rng.rand(s, s)).astype(np.float32))
    return np.ascontiguousarray(np.stack(out)[:, None])


def dense_text_maps(n: int, s: int, seed: int):
    """Post-processing stress maps: a grid of word-sized slanted boxes (about 50 per 640 x 640 frame)."""
    rng = np.random.RandomState(seed)
    out = np.empty((n, 1, s, s), np.float32)
    yy, xx = np.mgrid[0:s, 0:s]
    for i in range(n):
        m = np.zeros((s, s), bool)
        for gy in range(20, s - 40, 64):
            for gx in range(16, s - 90, 104):
                w, h = 60 + rng.randint(0, 30), 18 + rng.randint(0, 14)
                sl = rng.uniform(-0.15, 0.15)
                x0, y0 = gx + rng.randint(0, 8), gy + rng.randint(0, 8)
                m |= (xx >= x0) & (xx < x0 + w) & (yy >= y0 + sl * (xx - x0)) & (yy < y0 + h + sl * (xx - x0))
        out[i, 0] = np.where(m, 0.8 + 0.2 * rng.rand(s, s), 0.1 * rng.rand(s, s))
    return out

"""Fuzz of the device polygon chain against the host path (run on the GPU box): random maps - smoothed noise at several scales and
thresholds, rotated boxes, rings, speckle on top - in square sizes the device tracer takes, every batch through ocr_det_postprocess with the
chain on the device and entirely on the host; polygons and scores must be identical.   python3 tools/fuzz_chain.py [batches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
from tests import fixtures as FX
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
blob = W.pack_blob(W.make_det_weights(0))
host = capi.Detector(blob, 0, options="device_contours=0;device_unclip=0;post_threads=8")
dev = capi.Detector(blob, 0, options="device_contours=1;post_threads=2")
dev2 = capi.Detector(blob, 0, options="device_contours=0;device_unclip=2;post_threads=2")
rng = np.random.default_rng(2026)
params = capi.default_params(skip_degenerate=True)
tot = bad = 0
for b in range(nb):
    s = int(rng.choice([64, 128, 256, 320, 480, 640]))
    n = int(rng.integers(1, 5))
    maps = np.zeros((n, 1, s, s), np.float32)
    for i in range(n):
        kind = int(rng.integers(0, 4))
        m = rng.random((s, s))
        if kind == 0:      # smoothed noise
            for _ in range(int(rng.integers(3, 25))):
                m = (m + np.roll(m, 1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 0) + np.roll(m, -1, 1)) / 5
            m = (m - m.min()) / (m.max() - m.min() + 1e-9)
            m = np.clip((m - rng.uniform(0.35, 0.6)) * rng.uniform(3, 12) + 0.6, 0, 1)
        elif kind == 1:    # rotated boxes
            yy, xx = np.mgrid[0:s, 0:s]
            m = 0.1 * m
            for _ in range(int(rng.integers(3, 60))):
                cx, cy, w, h, th = rng.uniform(0, s), rng.uniform(0, s), rng.uniform(4, 90), rng.uniform(3, 30), rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
                v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                m = np.where((np.abs(u) < w / 2) & (np.abs(v) < h / 2), rng.uniform(0.62, 1.0), m)
        elif kind == 2:    # rings and blobs with holes
            yy, xx = np.mgrid[0:s, 0:s]
            m = 0.2 * m
            for _ in range(int(rng.integers(2, 20))):
                cx, cy, r0, r1 = rng.uniform(0, s), rng.uniform(0, s), rng.uniform(0, 20), rng.uniform(5, 50)
                d = np.hypot(xx - cx, yy - cy)
                m = np.where((d >= r0) & (d <= r0 + r1), rng.uniform(0.65, 0.95), m)
        else:              # text-like
            m = FX.dense_text_maps(1, s, int(rng.integers(0, 1 << 30)))[0, 0] if s >= 128 else m
        if rng.random() < 0.3:
            m = np.where(rng.random((s, s)) < 0.002, 0.9, m)   # speckle
        maps[i, 0] = m
    adj = np.stack([rng.uniform(0.4, 2.0, n), rng.uniform(0.4, 2.0, n)], 1)
    want = host.postprocess(maps, n, s, s, adj, capi.MEM_HOST, params)
    for name, d in (("chain", dev), ("unclip", dev2)):
        got = d.postprocess(maps, n, s, s, adj, capi.MEM_HOST, params)
        ok = got[0] == want[0] and all(np.array_equal(np.asarray(a), np.asarray(c)) for a, c in zip(got[1], want[1]))
        if not ok:
            bad += 1
            print(f"MISMATCH batch {b} ({name}) size {s} n {n}", flush=True)
    tot += sum(len(p) for p in want[0])
print(f"{nb} batches, {tot} polygons, {bad} mismatching batches")
sys.exit(1 if bad else 0)

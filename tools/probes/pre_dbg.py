import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
from oracle import preprocess_oracle as P
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
for (w,h,tw,th) in [(640,640,640,640),(3,5,64,32),(97,211,640,640)]:
    rng = np.random.RandomState(w * 7 + h)
    rgba = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    gray, ax, ay = det.preprocess_image(rgba, tw, th)
    og, _, _ = P.preprocess_image(rgba, tw, th)
    d = gray.astype(int) - og.astype(int)
    ys, xs = np.nonzero(d)
    print((w,h,tw,th), "mismatch", len(ys), "of", d.size, "max", np.abs(d).max() if len(ys) else 0)
    for y, x in list(zip(ys, xs))[:5]:
        print("   at", (y, x), "gpu", gray[y, x], "oracle", og[y, x])
    if (w,h)==(640,640):
        # identity resize: output should be luma of the input
        lum = P.to_luma(rgba)
        print("   identity: gpu==luma", np.array_equal(gray, lum), "oracle==luma", np.array_equal(og, lum))

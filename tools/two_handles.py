"""What can two half-batches on two streams buy?  Two detector handles, 16 frames each, on their own streams, against one handle with
32 frames (run on the GPU box).  free: both loops run freely (steps overlap across calls - an upper bound); step: both halves are
joined after every step (what ONE call that splits its batch can do); lag: stream 2 starts its half `lag` frames' worth later."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
blob = W.pack_blob(W.make_det_weights(0))
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
REPS = 20
def one(opt, n=32):
    det = capi.Detector(blob, 0, options=opt)
    for _ in range(3): det.forward_device(x.data_ptr(), n, 640, 640, prob.data_ptr(), 0, 0.6)
    det.synchronize()
    t = time.perf_counter()
    for _ in range(REPS): det.forward_device(x.data_ptr(), n, 640, 640, prob.data_ptr(), 0, 0.6)
    det.synchronize()
    ms = (time.perf_counter() - t) / REPS * 1e3
    det.close()
    return ms
def two(opt, mode, split=16):
    d = [capi.Detector(blob, 0, options=opt) for _ in range(2)]
    ns = [split, 32 - split]
    xs = [x[:split], x[split:]]
    ps = [prob[:split], prob[split:]]
    for _ in range(3):
        for k in range(2): d[k].forward_device(xs[k].data_ptr(), ns[k], 640, 640, ps[k].data_ptr(), 0, 0.6)
    for k in range(2): d[k].synchronize()
    t = time.perf_counter()
    for _ in range(REPS):
        for k in range(2): d[k].forward_device(xs[k].data_ptr(), ns[k], 640, 640, ps[k].data_ptr(), 0, 0.6)
        if mode == "step":
            for k in range(2): d[k].synchronize()
    for k in range(2): d[k].synchronize()
    ms = (time.perf_counter() - t) / REPS * 1e3
    for k in range(2): d[k].close()
    return ms
print("one handle, 32 frames:", round(one(None), 3), "ms;  16 frames:", round(one(None, 16), 3), "ms (x2 =", round(2 * one(None, 16), 3), ")")
for opt in (None, "w43_cus=192"):
    for mode in ("free", "step"):
        for split in (16, 12, 20):
            print(f"two handles {split}+{32 - split} frames, opt={opt}, {mode}: {two(opt, mode, split):.3f} ms per 32 frames", flush=True)
print("one handle, 32 frames:", round(one(None), 3))

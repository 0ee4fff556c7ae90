"""What can two half-batches on two streams buy?  Two detector handles, 16 frames each, free-running on their own streams, against one
handle with 32 frames (run on the GPU box).  `lag`: frames of an extra forward put in front of stream 2's loop (phase offset)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
blob = W.pack_blob(W.make_det_weights(0))
x = torch.from_numpy(W.synth_image_batch(1, 32, 640, 640)).cuda()
prob = torch.empty_like(x)
REPS = 20
def one(opt):
    det = capi.Detector(blob, 0, options=opt)
    for _ in range(3): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    det.synchronize()
    t = time.perf_counter()
    for _ in range(REPS): det.forward_device(x.data_ptr(), 32, 640, 640, prob.data_ptr(), 0, 0.6)
    det.synchronize()
    ms = (time.perf_counter() - t) / REPS * 1e3
    det.close()
    return ms
def two(opt, lag):
    d = [capi.Detector(blob, 0, options=opt) for _ in range(2)]
    xs = [x[:16], x[16:]]
    ps = [prob[:16], prob[16:]]
    for _ in range(3):
        for k in range(2): d[k].forward_device(xs[k].data_ptr(), 16, 640, 640, ps[k].data_ptr(), 0, 0.6)
    for k in range(2): d[k].synchronize()
    t = time.perf_counter()
    if lag: d[1].forward_device(xs[1].data_ptr(), lag, 640, 640, ps[1].data_ptr(), 0, 0.6)
    for _ in range(REPS):
        for k in range(2): d[k].forward_device(xs[k].data_ptr(), 16, 640, 640, ps[k].data_ptr(), 0, 0.6)
    for k in range(2): d[k].synchronize()
    ms = (time.perf_counter() - t) / REPS * 1e3
    for k in range(2): d[k].close()
    return ms
print("one handle, 32 frames:", round(one(None), 3), "ms per 32 frames")
print("one handle, 32 frames, w43_cus=128:", round(one("w43_cus=128"), 3))
for opt in (None, "w43_cus=128", "w43_cus=192"):
    for lag in (0, 4, 8, 12):
        print(f"two handles x 16 frames, opt={opt}, lag={lag}: {two(opt, lag):.3f} ms per 32 frames (lag forward included in the {REPS}-step window)", flush=True)

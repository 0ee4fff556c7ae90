"""Pipelined detect + post-process rate of a sequence of handles created one after the other in ONE process (diagnostic: does the rate
depend on what was created before?).  python3 tools/pipe_options.py "<opt>" "<opt>" ...  (16 pool threads, text pages)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
n, s, kk = 32, 640, 8
params = capi.default_params(skip_degenerate=True)
adj = np.ones((n, 2))
dev = torch.from_numpy(W.synth_text_pages(77, n, s, s)[0]).cuda()
pr = [torch.empty_like(dev), torch.empty_like(dev)]
blob = W.pack_blob(W.make_det_weights_text())
for opt in sys.argv[1:]:
    det = capi.Detector(blob, 0, options=opt if "post_threads" in opt else f"post_threads=16;{opt}")
    best = float("inf")
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(kk):
            det.detect_pipelined(dev.data_ptr(), n, s, s, pr[j & 1].data_ptr(), adj, params, convert=False)
        det.detect_pipelined(0, 0, 0, 0, 0, convert=False)
        torch.cuda.synchronize()
        if it: best = min(best, time.perf_counter() - t0)
    # forward only
    x = dev
    for _ in range(3): det.forward_device(x.data_ptr(), n, s, s, pr[0].data_ptr(), 0, 0.6)
    det.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): det.forward_device(x.data_ptr(), n, s, s, pr[0].data_ptr(), 0, 0.6)
    det.synchronize()
    fw = (time.perf_counter() - t0) / 10 * 1e3
    print(f"opt='{opt}': pipelined {n * kk / best:7.0f} frames/s, forward {fw:.3f} ms", flush=True)
    det.close()

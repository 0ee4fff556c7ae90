"""K pipelined detect calls (or forwards only) on dense pages, for a rocprofv3 --kernel-trace --stats comparison of per-kernel durations
with and without the polygon chain beside the forward.   python3 tools/pipeline_trace.py chain|forward [options]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
mode = sys.argv[1] if len(sys.argv) > 1 else "chain"
opts = sys.argv[2] if len(sys.argv) > 2 else "post_threads=2;device_contours=1;device_unclip=2"
n, s, K = 32, 640, 12
det = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0, options=opts)
stream = torch.cuda.Stream()
det.set_stream(stream.cuda_stream)
pages = torch.from_numpy(W.synth_text_pages(78, n, s, s, dense=True)[0]).cuda()
pr = [torch.empty_like(pages), torch.empty_like(pages)]
adj = np.ones((n, 2))
params = capi.default_params(skip_degenerate=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "chain":
        for j in range(K):
            det.detect_pipelined(pages.data_ptr(), n, s, s, pr[j & 1].data_ptr(), adj, params, convert=False)
        det.detect_pipelined(0, 0, 0, 0, 0, convert=False)
    else:
        for j in range(K):
            det.forward_device(pages.data_ptr(), n, s, s, pr[j & 1].data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(mode, opts, f"{n * K / dt:.1f} images/s, {dt / K * 1e3:.3f} ms per batch", flush=True)
det.close()

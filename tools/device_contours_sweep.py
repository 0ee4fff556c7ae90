#!/usr/bin/env python3
"""Pipelined detect + post-process (device-resident frames): where the polygon chain runs - host tracer + host unclip (round 4), host
tracer + device unclip, device tracer with the contours back on the host, the whole chain on the device - pool of 1 / 2 / 4 threads,
text-like and dense pages, f32 and bf16: python tools/device_contours_sweep.py [f32|bf16|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
n, s, kk = 32, 640, 8
params = capi.default_params(skip_degenerate=True)
adj = np.ones((n, 2))
pages = {"text": W.synth_text_pages(77, n, s, s)[0], "dense": W.synth_text_pages(78, n, s, s, dense=True)[0]}
dev = {k: torch.from_numpy(v).cuda() for k, v in pages.items()}
pr = [torch.empty_like(dev["text"]), torch.empty_like(dev["text"])]
blob = W.pack_blob(W.make_det_weights_text())
which = sys.argv[1] if len(sys.argv) > 1 else "both"
CONFIGS = (("host tracer, host unclip (r4)", "device_contours=0;device_unclip=0"), ("host tracer, device unclip", "device_contours=0"),
           ("device tracer, host DP", "device_contours=1;device_polygons=0"), ("device chain", "device_contours=1"),
           ("device chain, no head yield", "device_contours=1;head_cus_yield=0"))
# (no post_priority=0 configurations in this sequence: which hardware queue a stream gets depends on the streams the process made before -
#  DESIGN.md section 3.7 - and a handle with other stream priorities in the middle changes what the handles after it measure)
for prec in (("f32", "bf16") if which == "both" else (which,)):
    for threads in (1, 2, 4):
        for label, opt in CONFIGS:
            det = capi.Detector(blob, 0, options=f"post_threads={threads};{opt};precision={prec}")
            out = []
            for kind in ("text", "dense"):
                best, found = float("inf"), 0
                for it in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    got = 0
                    for j in range(kk):
                        r = det.detect_pipelined(dev[kind].data_ptr(), n, s, s, pr[j & 1].data_ptr(), adj, params, convert=False)
                        got += r[0] if r else 0
                    got += det.detect_pipelined(0, 0, 0, 0, 0, convert=False)[0]
                    torch.cuda.synchronize()
                    if it:
                        best, found = min(best, time.perf_counter() - t0), got
                out.append(f"{kind}: {n * kk / best:7.0f} frames/s ({found / (n * kk):.1f} polygons per page)")
            print(f"{prec} post_threads={threads} {label:32s}: " + "   ".join(out), flush=True)
            det.close()

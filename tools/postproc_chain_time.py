"""ocr_det_postprocess alone on detector output of dense / text pages (maps resident on the device): ms per batch of 32 for each placement of
the polygon chain.  Under rocprofv3 --kernel-trace --stats this gives the kernels' share.  python3 tools/postproc_chain_time.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, s = 32, 640
blob = W.pack_blob(W.make_det_weights_text())
params = capi.default_params(skip_degenerate=True)
adj = np.ones((n, 2))
maps = {}
d0 = capi.Detector(blob, 0)
for kind, dense in (("text", False), ("dense", True)):
    maps[kind] = torch.from_numpy(d0.forward_host(W.synth_text_pages(78, n, s, s, dense=dense)[0])).cuda()
d0.close()
for label, opt in (("host tracer, host unclip", "device_contours=0;device_unclip=0"), ("host tracer, device unclip", "device_contours=0"),
                   ("device tracer, host DP", "device_contours=1;device_polygons=0"), ("device chain", "device_contours=1")):
    det = capi.Detector(blob, 0, options=f"post_threads={threads};{opt}")
    out = []
    for kind in ("text", "dense"):
        m = maps[kind]
        for _ in range(2): det.postprocess_counts(m, n, s, s, adj, capi.MEM_DEVICE, params)
        t0 = time.perf_counter()
        for _ in range(5): c = det.postprocess_counts(m, n, s, s, adj, capi.MEM_DEVICE, params)
        out.append(f"{kind}: {(time.perf_counter() - t0) / 5 * 1e3:6.2f} ms ({c[0]} polygons)")
    print(f"threads={threads} {label:28s} " + "   ".join(out), flush=True)
    det.close()

"""How many candidates of real detector output does the device unclip settle?  (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()
det = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0)
L = capi.test_lib()
for kind, dense in (("text", False), ("dense", True)):
    pages = W.synth_text_pages(78, 8, 640, 640, dense=dense)[0]
    prob = det.forward_host(pages)
    polys = []
    for m in prob:
        polys += capi.host_contour_candidates((m[0] > 0.6).astype(np.uint8))
    xy = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32).reshape(-1) for p in polys]))
    cnt = np.array([len(p) for p in polys], np.int32)
    sc = np.full(len(polys), 0.9)
    stats = (C.c_int32 * 4)()
    st = np.zeros(len(polys), np.int32)
    ln = np.zeros(len(polys), np.int32)
    capi.check(L.ocr_test_unclip_compare(det._h, xy.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), len(polys), sc.ctypes.data_as(C.c_void_p),
                                         C.c_double(1.0), C.c_double(1.0), C.c_double(0.7), C.c_double(2.0), C.c_double(5.0), stats, st.ctypes.data_as(C.c_void_p),
                                         ln.ctypes.data_as(C.c_void_p), None))
    print(kind, "candidates", len(polys), "points per candidate", round(float(cnt.mean()), 1), "max", int(cnt.max()), "keep/host/drop/mismatch", list(stats))
    reasons = {}
    for k in np.nonzero(st == 2)[0]:
        reasons[int(-ln[k])] = reasons.get(int(-ln[k]), 0) + 1
    print("   handed back by reason (unclip.hip PUNT codes):", dict(sorted(reasons.items())))
    # why host?  concave vertices in the candidates
    conc = 0
    for p in polys:
        a = np.asarray(p, np.int64)
        n = len(a)
        area2 = sum(a[i][0] * a[(i + 1) % n][1] - a[(i + 1) % n][0] * a[i][1] for i in range(n))
        sgn = 1 if area2 > 0 else -1
        cr = [(a[i][0] - a[i - 1][0]) * (a[(i + 1) % n][1] - a[i][1]) - (a[i][1] - a[i - 1][1]) * (a[(i + 1) % n][0] - a[i][0]) for i in range(n)]
        conc += any(c * sgn < 0 for c in cr)
    print("   candidates with a concave vertex:", conc)

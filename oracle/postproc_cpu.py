"""ctypes front of oracle/postproc_cpu.cpp - the compiled CPU post-processing baseline (test infrastructure: only
tests/ and bench.py's cpu_baseline leg use it)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(_HERE, "libpostproc_cpu.so"))
        L.postproc_cpu.argtypes = ([C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_double] * 4 + [C.c_int, C.c_int] +
                                   [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int])
        _lib = L
    return _lib


def get_boxes_and_box_scores(pred, adjust_values, threads=1, skip_degenerate=False, counts_only=False,
                             thresh=0.6, box_thresh=0.7, min_size=5.0, unclip_ratio=2.0):
    """pred: N x 1 x H x W f32 -> (polygons per image as lists of (x, y), scores per image), like the Python oracle."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    n, _, h, w = pred.shape
    adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
    per = np.zeros(n, np.int32)
    tp, tv = C.c_int(0), C.c_int(0)
    cap_p, cap_v = (0, 0) if counts_only else (1 << 16, 1 << 20)
    lens = np.zeros(max(cap_p, 1), np.int32)
    xy = np.zeros(max(2 * cap_v, 1), np.uint32)
    sc = np.zeros(max(cap_p, 1), np.float64)
    rc = lib().postproc_cpu(pred.ctypes.data, n, h, w, adj.ctypes.data, thresh, box_thresh, min_size, unclip_ratio,
                            int(skip_degenerate), int(threads), per.ctypes.data, C.byref(tp), C.byref(tv),
                            None if counts_only else lens.ctypes.data, cap_p, None if counts_only else xy.ctypes.data, 2 * cap_v,
                            None if counts_only else sc.ctypes.data, cap_p)
    if rc:
        raise RuntimeError(f"postproc_cpu failed with {rc}")
    if counts_only:
        return tp.value, tv.value
    assert tp.value <= cap_p and tv.value <= cap_v
    polys, scores, ip, iv = [], [], 0, 0
    for b in range(n):
        pl, sl = [], []
        for _ in range(int(per[b])):
            L = int(lens[ip])
            pl.append([(int(xy[2 * (iv + k)]), int(xy[2 * (iv + k) + 1])) for k in range(L)])
            sl.append(float(sc[ip]))
            ip += 1
            iv += L
        polys.append(pl)
        scores.append(sl)
    return polys, scores

"""The device unclip (unclip.hip: score threshold, Clipper-style miter offset, the union's simple-ring case, min-size test,
round(p / adj) as u32 - /root/reference/src/text_detection/metrics.rs:100-123, src/polygon.rs:13-42) against the host geometry it
restates (postproc_geom.cpp, pinned to the reference's known answers) and against the Python oracle: whatever the kernel settles must
be bit for bit what the host does; what it hands back (non-simple rings, squared-off corners, short sides near min_size) is finished
by the host inside the same call - so ocr_det_postprocess gives the same polygons with device_unclip=1 (default) and 0."""
import ctypes as C

import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from tests import fixtures as FX
from oracle import postproc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    d = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
    yield d
    d.close()


def _compare(det, polys, scores, adj=(1.0, 1.0), box_thresh=0.7, ratio=2.0, min_size=5.0, want_polys=False):
    xy = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32).reshape(-1) for p in polys]))
    cnt = np.array([len(p) for p in polys], np.int32)
    sc = np.ascontiguousarray(scores, dtype=np.float64)
    stats = (C.c_int32 * 4)()
    st = np.zeros(len(polys), np.int32)
    ln = np.zeros(len(polys), np.int32)
    oxy = np.zeros(3 * int(cnt.sum()) * 2, np.uint32)
    L = capi.test_lib()
    capi.check(L.ocr_test_unclip_compare(det._h, xy.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), len(polys),
                                         sc.ctypes.data_as(C.c_void_p), C.c_double(adj[0]), C.c_double(adj[1]), C.c_double(box_thresh),
                                         C.c_double(ratio), C.c_double(min_size), stats, st.ctypes.data_as(C.c_void_p),
                                         ln.ctypes.data_as(C.c_void_p), oxy.ctypes.data_as(C.c_void_p)))
    res = {"keep": stats[0], "host": stats[1], "drop": stats[2], "mismatch": stats[3]}
    if want_polys:
        off = np.concatenate([[0], np.cumsum(cnt)])
        res["status"] = st
        res["polys"] = [[(int(oxy[6 * off[k] + 2 * i]), int(oxy[6 * off[k] + 2 * i + 1])) for i in range(ln[k])] for k in range(len(polys))]
    return res


def _word_boxes(rng, n):
    """slanted / rotated word-sized quadrilaterals and Douglas-Peucker-like 5..9-gons around them"""
    out = []
    for _ in range(n):
        cx, cy = rng.uniform(60, 580, 2)
        w, h = rng.uniform(20, 160), rng.uniform(6, 40)
        th = rng.uniform(-0.6, 0.6) if rng.random() < 0.8 else rng.uniform(0, np.pi)
        k = int(rng.integers(4, 10))
        # points on the rectangle's outline at sorted arc positions, jittered: convex or mildly concave rings
        t = np.sort(rng.uniform(0, 4, k))
        pts = []
        for u in t:
            s, f = int(u), u - int(u)
            corners = [(-w / 2, -h / 2), (w / 2, -h / 2), (w / 2, h / 2), (-w / 2, h / 2)]
            (x0, y0), (x1, y1) = corners[s], corners[(s + 1) % 4]
            x, y = x0 + f * (x1 - x0) + rng.normal(0, 0.8), y0 + f * (y1 - y0) + rng.normal(0, 0.8)
            pts.append((int(round(cx + x * np.cos(th) - y * np.sin(th))), int(round(cy + x * np.sin(th) + y * np.cos(th)))))
        out.append(pts)
    return out


def _odd_shapes(rng, n):
    """concave arrows / L shapes (the miter offset loops: non-simple rings), needles (squared-off corners), specks near min_size,
    axis-aligned rectangles with exact half-pixel offsets, degenerate and repeated points"""
    out = []
    for i in range(n):
        kind = i % 7
        x0, y0 = int(rng.integers(40, 500)), int(rng.integers(40, 500))
        if kind == 0:    # L
            a, b, c = int(rng.integers(20, 90)), int(rng.integers(20, 90)), int(rng.integers(4, 15))
            out.append([(x0, y0), (x0 + a, y0), (x0 + a, y0 + c), (x0 + c, y0 + c), (x0 + c, y0 + b), (x0, y0 + b)])
        elif kind == 1:  # arrow / chevron
            a, b = int(rng.integers(30, 120)), int(rng.integers(10, 40))
            out.append([(x0, y0), (x0 + a, y0 + b), (x0 + 2 * a, y0), (x0 + a, y0 + b // 3)])
        elif kind == 2:  # needle: a very sharp convex corner
            a = int(rng.integers(40, 200))
            out.append([(x0, y0), (x0 + a, y0 + int(rng.integers(1, 4))), (x0 + a, y0 + int(rng.integers(4, 9))), (x0 + 3, y0 + int(rng.integers(9, 14)))])
        elif kind == 3:  # specks around min_size
            a, b = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            out.append([(x0, y0), (x0 + a, y0), (x0 + a, y0 + b), (x0, y0 + b)])
        elif kind == 4:  # axis-aligned rectangles (d = w h / (w + h): exact, often k + 0.5)
            a, b = int(rng.integers(3, 120)), int(rng.integers(3, 60))
            out.append([(x0, y0), (x0 + a, y0), (x0 + a, y0 + b), (x0, y0 + b)][::(1 if i % 2 else -1)])
        elif kind == 5:  # repeated and collinear points
            a, b = int(rng.integers(10, 80)), int(rng.integers(10, 40))
            out.append([(x0, y0), (x0, y0), (x0 + a // 2, y0), (x0 + a, y0), (x0 + a, y0 + b), (x0, y0 + b), (x0, y0)])
        else:            # random polygon (self-intersections allowed)
            k = int(rng.integers(4, 12))
            out.append([(int(x0 + rng.integers(0, 80)), int(y0 + rng.integers(0, 80))) for _ in range(k)])
    return out


def test_hypot_port_is_libm_hypot():
    """the perimeter on the device adds up glibc's hypot, not sqrt(dx^2 + dy^2): the restated kernel against std::hypot (host code, no GPU work)"""
    L = capi.test_lib()
    L.ocr_test_hypot_port_mismatches.restype = C.c_longlong
    assert L.ocr_test_hypot_port_mismatches(1500) == 0


def _real_candidates(maps):
    """what the kernel sees in use: contours of thresholded maps through the product's Douglas-Peucker (>= 4 points)"""
    out = []
    for m in maps:
        out += capi.host_contour_candidates((m[0] > 0.6).astype(np.uint8))
    return out


def test_real_candidates_are_settled_on_the_device_and_equal_the_host(det):
    """candidates of dense maps (word boxes) and text-like maps (the reference's shrunk ground-truth polygons of curved text: concave -
    every concave vertex makes the miter offset loop, and such rings are the host union's): all of what the device settles equals the
    host, and word boxes ARE settled there"""
    rng = np.random.default_rng(20)
    for name, maps, most in (("dense", FX.dense_text_maps(6, 640, 31), 0.9), ("text-like", FX.text_like_maps(8, 800, 32), 0.2)):
        polys = _real_candidates(maps)
        assert len(polys) > 20, name
        scores = rng.uniform(0.65, 1.0, len(polys))
        for adj in ((1.0, 1.0), (0.8, 0.53125)):
            st = _compare(det, polys, scores, adj=adj)
            assert st["mismatch"] == 0, (name, st)
            assert st["keep"] >= most * (st["keep"] + st["host"]), (name, st)


def test_word_boxes_fuzz_equals_the_host(det):
    rng = np.random.default_rng(21)
    polys = _word_boxes(rng, 6000)
    scores = rng.uniform(0.5, 1.0, len(polys))
    for adj in ((1.0, 1.0), (0.8, 0.53125), (1.7, 2.0)):
        st = _compare(det, polys, scores, adj=adj)
        assert st["mismatch"] == 0, st
        assert st["keep"] + st["host"] + st["drop"] == len(polys)
        assert st["keep"] > 300 and st["host"] > 300, st   # (points jittered along an edge make concave vertices: many offset rings loop)


def test_odd_shapes_never_disagree_with_the_host(det):
    rng = np.random.default_rng(22)
    polys = _odd_shapes(rng, 7000)
    scores = rng.uniform(0.6, 1.0, len(polys))
    scores[::50] = np.nan   # 0 / 0 box score: passes the threshold, as in the reference
    for kw in ({}, {"min_size": 3.0}, {"ratio": 1.5, "box_thresh": 0.8}, {"adj": (0.37, 1.9)}):
        st = _compare(det, polys, scores, **kw)
        assert st["mismatch"] == 0, (kw, st)
        assert st["keep"] > 1000 and st["host"] > 100, st   # both paths exercised


def test_kept_polygons_equal_the_python_oracle(det):
    """... and the oracle itself (oracle/postproc_oracle.py: expand_polygon, min-area box, round(p / adj) as u32) on a sample the
    pure-Python code finishes in seconds: a polygon the device keeps is the oracle's polygon, one it drops fails the oracle's score test"""
    rng = np.random.default_rng(23)
    polys = _word_boxes(rng, 260) + _odd_shapes(rng, 140)
    scores = rng.uniform(0.6, 1.0, len(polys))
    adj = (0.8, 0.53125)
    res = _compare(det, polys, scores, adj=adj, want_polys=True)
    assert res["mismatch"] == 0
    kept = 0
    for p, sc, st, got in zip(polys, scores, res["status"], res["polys"]):
        if st == 0:
            assert 0.7 > sc
            continue
        if st != 1:
            continue
        exp = O.expand_polygon(p, 2.0)
        assert exp, p
        _, sside = O.get_min_area_bounding_box(exp)
        assert not sside < 5.0, p
        assert got == [(O._as_u32(O._round_half_away(x / adj[0])), O._as_u32(O._round_half_away(y / adj[1]))) for x, y in exp], p
        kept += 1
    assert kept > 50


def _post(det, maps, adj, **kw):
    p = capi.default_params(skip_degenerate=True)
    for k, v in kw.items():
        setattr(p, k, v)
    return det.postprocess(maps, maps.shape[0], maps.shape[2], maps.shape[3], adj, capi.MEM_HOST, p)


def test_postprocess_is_the_same_with_and_without_the_device_unclip():
    blob = W.pack_blob(W.make_det_weights(0))
    dev = capi.Detector(blob, 0, options="device_unclip=2")
    host = capi.Detector(blob, 0, options="device_unclip=0")
    rng = np.random.default_rng(5)
    noise = (rng.random((2, 1, 320, 320)) * 0.95).astype(np.float32)
    m = rng.random((2, 1, 256, 256))
    for _ in range(30):
        m = (m + np.roll(m, 1, 2) + np.roll(m, 1, 3) + np.roll(m, -1, 2) + np.roll(m, -1, 3)) / 5
    smooth = ((m - m.min()) / (m.max() - m.min())).astype(np.float32)
    smooth = np.clip((smooth - 0.5) * 6 + 0.55, 0, 1).astype(np.float32)
    total = 0
    for name, maps in (("text", FX.text_like_maps(4, 640, 1)), ("dense", FX.dense_text_maps(4, 640, 2)), ("noise", noise), ("smooth", smooth),
                       ("800", FX.text_like_maps(1, 800, 9))):
        n = maps.shape[0]
        for adj in (np.ones((n, 2)), np.tile([[0.8, 0.53125]], (n, 1))):
            a = _post(dev, maps, adj)
            b = _post(host, maps, adj)
            assert a[0] == b[0], name
            assert a[1] == b[1], name
            total += sum(len(x) for x in a[0])
    assert total > 300
    dev.close()
    host.close()


def test_reference_known_answer_through_the_device_unclip(golden_dir):
    """metrics.rs:510-646 (img55, adj 1 and 2): polygons 2 and 4 are concave - their offset rings loop and go back to the host -,
    1 and 3 are settled on the device; the batch result is the reference's"""
    from PIL import Image
    from tests import kat_postproc as K
    import os
    img = np.array(Image.open(os.path.join(golden_dir, "gt_shrinked_img55.png")).convert("L"))
    pred = (img.astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)
    d = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options="device_unclip=2")
    polys, scores = d.postprocess(pred, 1, 800, 800, np.array([[1.0, 1.0]]))
    assert polys[0] == K.IMG55_POLYS_ADJ1 and scores[0] == K.IMG55_SCORES
    polys, scores = d.postprocess(pred, 1, 800, 800, np.array([[2.0, 2.0]]))
    assert polys[0] == K.IMG55_POLYS_ADJ2
    d.close()


def _random_maps(rng, n, s):
    """smoothed noise at several scales, rotated boxes, rings with holes, dense text, speckle: tools/fuzz_chain.py's generator"""
    maps = np.zeros((n, 1, s, s), np.float32)
    for i in range(n):
        kind = int(rng.integers(0, 4))
        m = rng.random((s, s))
        if kind == 0:
            for _ in range(int(rng.integers(3, 25))):
                m = (m + np.roll(m, 1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 0) + np.roll(m, -1, 1)) / 5
            m = (m - m.min()) / (m.max() - m.min() + 1e-9)
            m = np.clip((m - rng.uniform(0.35, 0.6)) * rng.uniform(3, 12) + 0.6, 0, 1)
        elif kind == 1:
            yy, xx = np.mgrid[0:s, 0:s]
            m = 0.1 * m
            for _ in range(int(rng.integers(3, 60))):
                cx, cy, w, h, th = rng.uniform(0, s), rng.uniform(0, s), rng.uniform(4, 90), rng.uniform(3, 30), rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
                v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                m = np.where((np.abs(u) < w / 2) & (np.abs(v) < h / 2), rng.uniform(0.62, 1.0), m)
        elif kind == 2:
            yy, xx = np.mgrid[0:s, 0:s]
            m = 0.2 * m
            for _ in range(int(rng.integers(2, 20))):
                cx, cy, r0, r1 = rng.uniform(0, s), rng.uniform(0, s), rng.uniform(0, 20), rng.uniform(5, 50)
                d = np.hypot(xx - cx, yy - cy)
                m = np.where((d >= r0) & (d <= r0 + r1), rng.uniform(0.65, 0.95), m)
        elif s >= 128:
            m = FX.dense_text_maps(1, s, int(rng.integers(0, 1 << 30)))[0, 0]
        if rng.random() < 0.3:
            m = np.where(rng.random((s, s)) < 0.002, 0.9, m)
        maps[i, 0] = m
    return maps


def test_random_maps_give_the_same_polygons_wherever_the_chain_runs():
    """30 random batches (tools/fuzz_chain.py runs 120: 16 131 polygons, no mismatch): host path, device unclip behind the host tracer, the whole
    chain on the device - identical polygon lists and scores, random adjust values"""
    blob = W.pack_blob(W.make_det_weights(0))
    host = capi.Detector(blob, 0, options="device_contours=0;device_unclip=0")
    chain = capi.Detector(blob, 0, options="device_contours=1")
    unclip = capi.Detector(blob, 0, options="device_contours=0;device_unclip=2")
    rng = np.random.default_rng(77)
    params = capi.default_params(skip_degenerate=True)
    total = 0
    for b in range(30):
        s = int(rng.choice([64, 128, 256, 320, 640]))
        n = int(rng.integers(1, 4))
        maps = _random_maps(rng, n, s)
        adj = np.stack([rng.uniform(0.4, 2.0, n), rng.uniform(0.4, 2.0, n)], 1)
        want = host.postprocess(maps, n, s, s, adj, capi.MEM_HOST, params)
        for name, d in (("chain", chain), ("unclip", unclip)):
            got = d.postprocess(maps, n, s, s, adj, capi.MEM_HOST, params)
            assert got[0] == want[0], (b, name, s, n)
            assert all(np.array_equal(np.asarray(a), np.asarray(c)) for a, c in zip(got[1], want[1])), (b, name)
        total += sum(len(p) for p in want[0])
    assert total > 1000
    for d in (host, chain, unclip):
        d.close()

"""The polygon chain ON THE DEVICE (contours.hip -> candidates.hip -> box_score.hip -> unclip.hip) held to the ORACLE directly - not
through the product's host path - and to the reference's own known answers at the reference's own frame size:

    /root/reference/src/text_detection/mod.rs:20-21     default frame 800 x 800 (every fixture, both end-to-end KATs)
    /root/reference/src/text_detection/metrics.rs:78-98  find_contours, arc_length, approximate_polygon_dp, >= 4 points
    metrics.rs:510-646                                    get_polygons_from_bitmap on gt_shrinked_img55.png, adj 1 and 2

`ocr_det_post_stats` says where each step ran: these tests assert that the image was NOT handed back to the host."""
import os

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import text_detection as td
from ocr_rs_amd import weights as W
from oracle import postproc_oracle as O
from tests import fixtures as FX
from tests import kat_postproc as K

pytestmark = pytest.mark.gpu

CHAIN = "device_contours=1;device_unclip=2"          # every step on the device, whatever the pool size
FIXTURES = ["gt_shrinked_img55.png", "gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]


def _img(golden_dir, name):
    return np.array(Image.open(os.path.join(golden_dir, name)).convert("L"))


@pytest.fixture(scope="module")
def blob():
    return W.pack_blob(W.make_det_weights(0))


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_fixtures_800_through_the_device_tracer(golden_dir, name):
    """the reference's 800 x 800 label maps: the device tracer takes them (status 0) and returns the ORACLE's contours, point for point"""
    bm = (_img(golden_dir, name) > 0).astype(np.uint8)
    assert bm.shape == (800, 800)
    got, status = capi.device_contours(bm)
    assert status == 0
    want = [[(int(x), int(y)) for x, y in c] for c in O.find_contours(bm * 255)]
    assert [[tuple(q) for q in c] for c in got] == want
    if name == "gt_shrinked_img55.png":
        assert [len(c) for c in got] == [239, 463, 196, 505]


def _oracle_candidates(contours):
    """metrics.rs:86-98 with the oracle's arc_length / approximate_polygon_dp"""
    out = []
    for c in contours:
        eps = 0.01 * O.arc_length(c, True)
        if eps == 0.0:
            eps = 0.01
        pts = O.approximate_polygon_dp(c, eps, True)
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts.pop()
        if len(pts) >= 4:
            out.append([(int(x), int(y)) for x, y in pts])
    return out


def _box(pts, h, w):
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    cl = lambda v, hi: min(max(v, 0), hi - 1)   # noqa: E731
    x0, x1, y0, y1 = cl(min(xs), h), cl(max(xs), h), cl(min(ys), w), cl(max(ys), w)   # x by H, y by W: metrics.rs:151-166
    return (x0, y0, x1 - x0 + 1, y1 - y0 + 1)


def test_device_douglas_peucker_equals_the_oracle(golden_dir):
    """dp_kernel / cand_count / cand_fill (candidates.hip) against O.arc_length + O.approximate_polygon_dp: on the oracle's contours of
    the four reference fixtures (800 x 800), of text-like and dense pages, and of noise and blobs - hundreds of short contours, straight
    runs (ties for the farthest point), contours of fewer than four points."""
    rng = np.random.default_rng(17)
    maps = [(n, (_img(golden_dir, n) > 0).astype(np.uint8)) for n in FIXTURES]
    maps.append(("text-like 640", (FX.text_like_maps(1, 640, 5)[0, 0] > 0.6).astype(np.uint8)))
    maps.append(("dense 320", (FX.dense_text_maps(1, 320, 6)[0, 0] > 0.6).astype(np.uint8)))
    for h, w, p in ((96, 96, 0.5), (128, 160, 0.35), (64, 224, 0.65)):
        maps.append((f"noise {h}x{w}", (rng.random((h, w)) < p).astype(np.uint8)))
    m = rng.random((192, 192))
    for _ in range(30):
        m = (m + np.roll(m, 1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 0) + np.roll(m, -1, 1)) / 5
    maps.append(("blobs", (m > np.median(m)).astype(np.uint8)))
    ring = np.zeros((96, 128), np.uint8)
    ring[10:40, 10:50] = 1; ring[11:39, 11:49] = 0; ring[50:90, 20:100] = 1; ring[55:85, 25:95] = 0; ring[5, 60:120] = 1; ring[5:45, 120] = 1
    maps.append(("rings and lines", ring))
    total = 0
    for name, bm in maps:
        h, w = bm.shape
        contours = [[(int(x), int(y)) for x, y in c] for c in O.find_contours(bm * 255)]
        want = _oracle_candidates(contours)
        got, boxes = capi.device_candidates(contours, h, w)
        assert got == want, name
        if h == w:
            assert boxes == [_box(c, h, w) for c in want], name
        total += len(want)
    assert total > 100


@pytest.mark.parametrize("adj,expected", [((1.0, 1.0), K.IMG55_POLYS_ADJ1), ((2.0, 2.0), K.IMG55_POLYS_ADJ2)])
def test_reference_kat_through_the_device_chain(blob, golden_dir, adj, expected):
    """metrics.rs:510-646 at the reference's 800 x 800 with EVERY step on the device: trace, Douglas-Peucker, box scores, unclip (two of
    the four polygons are concave), adjustment - nothing handed back."""
    det = capi.Detector(blob, 0, options=CHAIN)
    pred = (_img(golden_dir, "gt_shrinked_img55.png").astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)
    polys, scores = det.postprocess(pred, 1, 800, 800, np.array([adj]))
    st = det.post_stats()
    det.close()
    assert polys[0] == expected
    assert scores[0] == K.IMG55_SCORES
    assert st["images_device_chain"] == 1 and st["images_host_traced"] == 0, st
    assert st["candidates_device"] == 4 and st["candidates_host"] == 0, st


@pytest.mark.parametrize("adj,expected", [((1.0, 1.0), K.IMG55_POLYS_ADJ1), ((2.0, 2.0), K.IMG55_POLYS_ADJ2)])
def test_reference_style_api_through_the_device_chain(blob, golden_dir, adj, expected):
    """the same through the reference-named mirror (text_detection.get_boxes_and_box_scores), as tests/test_reference_style_gpu.py:32,38"""
    net = td.resnet18(blob, 0, options=CHAIN)
    pred = (_img(golden_dir, "gt_shrinked_img55.png").astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)
    res = td.get_boxes_and_box_scores(net, pred, np.array([list(adj)]))
    st = net.handle.post_stats()
    net.close()
    assert res.polygons[0] == expected
    assert res.scores[0] == K.IMG55_SCORES
    assert st["images_device_chain"] == 1 and st["candidates_host"] == 0, st


def test_all_reference_fixtures_device_chain_equals_the_oracle(blob, golden_dir):
    """the four 800 x 800 fixtures as one batch, non-trivial adjust values: device chain == O.get_boxes_and_box_scores, all on the device"""
    pred = np.stack([(_img(golden_dir, n).astype(np.float64) / 255.0).astype(np.float32)[None] for n in FIXTURES])
    adj = np.array([[800 / 300, 533 / 200], [1.0, 1.0], [2.0, 2.0], [800 / 240, 600 / 180]])
    want_p, want_s = O.get_boxes_and_box_scores(pred, adj)
    det = capi.Detector(blob, 0, options=CHAIN)
    polys, scores = det.postprocess(pred, 4, 800, 800, adj)
    st = det.post_stats()
    det.close()
    assert polys == want_p
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(scores, want_s))
    # (a candidate the unclip kernel does not settle - a short side within 3 px of min_size, a ring that touches itself - is finished by
    # the host inside the same call: one of the twelve here)
    assert st["images_device_chain"] == 4 and st["images_host_traced"] == 0 and st["candidates_host"] <= 2, st
    assert st["candidates_device"] + st["candidates_host"] == 12 and sum(len(p) for p in polys) >= 10


def test_frames_to_polygons_at_the_reference_frame_size_against_the_oracle_chain():
    """The reference's default frame (text_detection/mod.rs:20-21: 800 x 800) from FRAMES to polygons with the forward and every step of
    the chain on the device, against the oracle chain on its own map (ATen CPU -> Python restatement): dense pages (about 100 words each),
    pages without a binarisation flip must give identical lists."""
    from oracle import torch_ref as T
    n, s = 3, 800
    det_w = W.make_det_weights_text()
    frames, boxes = W.synth_text_pages(2031, n, s, s, dense=True)
    det = capi.Detector(W.pack_blob(det_w), 0, options=CHAIN)
    prob = det.forward_host(frames)
    adj = np.array([[800 / 300, 533 / 200], [1.0, 1.0], [1.25, 0.8]])
    polys, scores = det.postprocess(prob, n, s, s, adj, capi.MEM_HOST, capi.default_params(skip_degenerate=True))
    st = det.post_stats()
    det.close()
    ref_prob = np.concatenate([T.det_forward(det_w, frames[i:i + 1]) for i in range(n)])
    ref_polys, ref_scores = O.get_boxes_and_box_scores(ref_prob, adj, skip_degenerate=True)
    assert float(np.abs(prob - ref_prob).max()) < 1e-4
    assert st["images_device_chain"] == n and st["images_host_traced"] == 0, st
    flips = ((prob > np.float32(0.6)) != (ref_prob > np.float32(0.6))).reshape(n, -1).sum(axis=1)
    assert sum(len(p) for p in ref_polys) > 60 * n
    same = 0
    for i in range(n):
        if flips[i] == 0:
            assert polys[i] == ref_polys[i], f"page {i}"
            assert np.allclose(scores[i], ref_scores[i], rtol=0, atol=1e-6)
            same += 1
    assert same >= n - 1

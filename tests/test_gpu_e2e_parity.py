"""End to end against the ORACLE CHAIN, each side on its own probability map (north_star: "outputs match the reference
libtorch-CPU path on identical inputs - bit-exact polygon vertex indices and label sequences, fp32 probability maps within 1e-4"):

    GPU     ocr_det_forward -> ocr_det_postprocess -> ocr_extract_crops -> ocr_rec_classify        (default engine, f32)
    oracle  T.det_forward   -> O.get_boxes_and_box_scores -> CR.extract_crops -> T.rec_classify     (ATen CPU + Python restatement)

what a caller of /root/reference/src/text_detection/mod.rs:52-67 and char_recognition/mod.rs:53-56 would see for a
BASELINE configs[1]-sized batch (32 pages of 640 x 640).  Every other polygon assertion of the suite runs both sides on the
SAME (GPU) map; here a pixel within rounding of the 0.6 threshold may flip the bitmap, so the test counts the flips and what
they do: pages without a flip must agree exactly, and the batch as a whole almost everywhere."""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import crop_oracle as CR
from oracle import postproc_oracle as O
from oracle import torch_ref as T
from tests.test_gpu_config4 import run_pipeline

pytestmark = pytest.mark.gpu
S, N = 640, 32


# where the polygon chain runs: the engine's default for this box, everything on the host pool, everything on the device
CHAINS = {"default": None, "host chain": "device_contours=0;device_unclip=0", "device chain": "device_contours=1;device_unclip=2"}


@pytest.fixture(scope="module")
def oracle_32_pages():
    """the oracle chain on ITS OWN map, once for every placement of the product's chain"""
    det_w, rec_w = W.make_det_weights_text(), W.make_rec_weights(0)
    frames, boxes = W.synth_text_pages(2026, N, S, S)
    ref_prob = np.concatenate([T.det_forward(det_w, frames[i:i + 8]) for i in range(0, N, 8)])
    adj = np.ones((N, 2))
    ref_polys, ref_scores = O.get_boxes_and_box_scores(ref_prob, adj, skip_degenerate=True)
    ref_crops = CR.extract_crops(frames, ref_polys, adj)
    ref_logits = T.rec_forward(rec_w, ref_crops)
    ref_labels, _ = T.rec_classify(ref_logits)
    return det_w, rec_w, frames, boxes, ref_prob, ref_polys, ref_scores, ref_crops, ref_logits, ref_labels


@pytest.mark.parametrize("chain", list(CHAINS), ids=list(CHAINS))
def test_32_pages_frames_to_polygons_to_labels_against_the_oracle_chain(oracle_32_pages, chain):
    det_w, rec_w, frames, boxes, ref_prob, ref_polys, ref_scores, ref_crops, ref_logits, ref_labels = oracle_32_pages
    det = capi.Detector(W.pack_blob(det_w), 0, options=CHAINS[chain])
    rec = capi.Recognizer(W.pack_blob(rec_w), 0)
    prob, polys, scores, crops, labels = run_pipeline(det, rec, frames, capi.PRECISION_F32)
    st = det.post_stats()
    det.close()
    rec.close()
    if chain == "device chain":     # ... and it did run there: no page handed back to the host tracer, (almost) no polygon to the host unclip
        assert st["images_device_chain"] == N and st["images_host_traced"] == 0, st
        assert st["candidates_host"] <= 0.02 * st["candidates_device"], st
    if chain == "host chain":
        assert st["images_device_traced"] == 0 and st["candidates_device"] == 0, st
    adj = np.ones((N, 2))

    # ---- maps: 1e-4, and how many pixels sit on the other side of the threshold
    d = np.abs(prob - ref_prob)
    flips_px = (prob > np.float32(0.6)) != (ref_prob > np.float32(0.6))
    flips = flips_px.reshape(N, -1).sum(axis=1)
    near = int((np.abs(ref_prob - np.float32(0.6)) < 1e-5).sum())
    print(f"e2e {N} pages: max|dprob| {d.max():.3e}, {int(flips.sum())} binarisation flips of {d.size} pixels on {int((flips > 0).sum())} pages; "
          f"{near} oracle pixels within 1e-5 of the threshold")
    assert d.max() < 1e-4
    assert flips.sum() <= 1e-6 * d.size

    # ---- polygons, scores, crops, labels page by page
    assert [len(p) for p in ref_polys] == [len(b) for b in boxes]       # the oracle finds every word box
    k_gpu = np.concatenate([[0], np.cumsum([len(p) for p in polys])])
    k_ref = np.concatenate([[0], np.cumsum([len(p) for p in ref_polys])])
    srt = np.sort(ref_logits, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3
    top2 = np.argsort(ref_logits, axis=1)[:, -2:]
    identical_pages, undecided = 0, 0
    for i in range(N):
        same = polys[i] == ref_polys[i]
        if flips[i] == 0:
            assert same, f"page {i}: no pixel flipped, polygon lists differ"
        if not same:
            continue
        identical_pages += 1
        assert np.allclose(scores[i], ref_scores[i], rtol=0, atol=1e-6)
        g, r = slice(k_gpu[i], k_gpu[i + 1]), slice(k_ref[i], k_ref[i + 1])
        assert np.array_equal(crops[g], ref_crops[r])                    # same polygons, same frames: same crops bit for bit
        for a, b in zip(range(g.start, g.stop), range(r.start, r.stop)):
            if decided[b]:
                assert labels[a] == ref_labels[b]
            else:
                undecided += 1
                assert labels[a] in top2[b]
    n_poly = int(k_ref[-1])
    print(f"e2e {N} pages: {identical_pages} pages with identical polygon lists, {n_poly} oracle polygons, {undecided} near-tie crops")
    assert identical_pages >= N - 2
    assert n_poly > 10 * N


@pytest.fixture(scope="module")
def oracle_soft_pages():
    det_w = W.make_det_weights_text(gain=0.01, tau=84.5)
    frames, boxes = W.synth_text_pages(2027, N, S, S)
    ref_prob = np.concatenate([T.det_forward(det_w, frames[i:i + 8]) for i in range(0, N, 8)])
    ref_polys, ref_scores = O.get_boxes_and_box_scores(ref_prob, np.ones((N, 2)), skip_degenerate=True)
    return det_w, frames, ref_prob, ref_polys, ref_scores


@pytest.mark.parametrize("chain", list(CHAINS), ids=list(CHAINS))
def test_maps_that_straddle_the_threshold_differ_only_where_a_pixel_flipped(oracle_soft_pages, chain):
    """The same two chains on SOFT maps: the text-following weights with a flat sigmoid (gain 0.01 instead of 0.06) leave more than 1e-4 of
    the oracle's pixels within 1e-3 of the 0.6 threshold (/root/reference/src/text_detection/metrics.rs:38,129-131), so float rounding
    does move pixels across it.  Counted and explained: pages without a flipped pixel have identical polygon lists; on every other page
    each polygon that one side has and the other has not holds a flipped pixel inside its bounding box."""
    det_w, frames, ref_prob, ref_polys, ref_scores = oracle_soft_pages
    det = capi.Detector(W.pack_blob(det_w), 0, options=CHAINS[chain])
    prob = det.forward_host(frames)
    adj = np.ones((N, 2))
    polys, scores = det.postprocess(prob, N, S, S, adj, capi.MEM_HOST, capi.default_params(skip_degenerate=True))
    st = det.post_stats()
    det.close()
    if chain == "device chain":
        assert st["images_device_chain"] == N and st["images_host_traced"] == 0, st

    d = np.abs(prob - ref_prob)
    near = float((np.abs(ref_prob - np.float32(0.6)) < 1e-3).mean())
    flips_px = (prob > np.float32(0.6)) != (ref_prob > np.float32(0.6))
    flips = flips_px.reshape(N, -1).sum(axis=1)
    assert d.max() < 1e-4
    assert near >= 1e-4, near                       # the maps do straddle the threshold ...
    assert sum(len(p) for p in ref_polys) > 10 * N  # ... and still carry the words
    differing, explained = 0, 0
    for i in range(N):
        if flips[i] == 0:
            assert polys[i] == ref_polys[i], f"page {i}: no pixel flipped, polygon lists differ"
            assert np.allclose(scores[i], ref_scores[i], rtol=0, atol=1e-6)
            continue
        a, b = {tuple(map(tuple, q)) for q in polys[i]}, {tuple(map(tuple, q)) for q in ref_polys[i]}
        ys, xs = np.nonzero(flips_px[i, 0])
        for q in a ^ b:
            differing += 1
            qa = np.asarray(q)
            x0, y0, x1, y1 = qa[:, 0].min() - 2, qa[:, 1].min() - 2, qa[:, 0].max() + 2, qa[:, 1].max() + 2
            inside = ((xs >= x0) & (xs <= x1) & (ys >= y0) & (ys <= y1)).any()
            assert inside, f"page {i}: polygon {q[:3]}... differs between the chains and no flipped pixel lies in its box"
            explained += 1
    print(f"soft maps, {N} pages: {near:.2e} of the oracle's pixels within 1e-3 of the threshold, max|dprob| {d.max():.3e}, {int(flips.sum())} flipped pixels on "
          f"{int((flips > 0).sum())} pages, {differing} polygons differ, all {explained} with a flipped pixel in their box")

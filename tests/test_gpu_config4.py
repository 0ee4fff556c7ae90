"""BASELINE configs[4]: end-to-end detect -> polygons -> crops -> recognise with the detector in the opt-in bf16
precision (bf16 operands on the matrix cores, f32 accumulate), 640 x 640 pages.

The reference is f32 only and ships no weights, so what can be held is
  (1) the f32 pipeline against the oracles on the same pages (parity proper),
  (2) bf16 against f32 on the same pages: polygon / label mismatch COUNTS, bounded,
  (3) on the full 128 pages: size-independent properties of the bf16 pipeline.
Random weights give noise maps on which polygon lists of two precisions cannot be compared; the pages run on
weights.make_det_weights_text (a signal path that follows the ink + 2 % of the random weights everywhere)."""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import crop_oracle as CR
from oracle import postproc_oracle as O
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu
S = 640


@pytest.fixture(scope="module")
def det_w():
    return W.make_det_weights_text()


@pytest.fixture(scope="module")
def rec_w():
    return W.make_rec_weights(0)


def run_pipeline(det, rec, frames, precision):
    """Device-resident pass: forward (+ fused binarize) -> get_boxes_and_box_scores -> crops -> classify."""
    import torch
    n = frames.shape[0]
    det.set_precision(precision)
    x = torch.from_numpy(frames).cuda()
    prob = torch.empty_like(x)
    torch.cuda.synchronize()
    det.forward_device(x.data_ptr(), n, S, S, prob.data_ptr())
    det.synchronize()
    hold = {}

    def alloc(p):
        hold["crops"] = torch.empty((p, 784), dtype=torch.float32, device="cuda")
        return hold["crops"].data_ptr()

    adj = np.ones((n, 2))
    polys, scores, npoly = det.postprocess_and_crops_device(prob.data_ptr(), x.data_ptr(), n, S, S, adj, alloc,
                                                            capi.default_params(skip_degenerate=True))
    labels = np.zeros(0, np.int32)
    crops = np.zeros((0, 784), np.float32)
    if npoly:
        lab = torch.empty(npoly, dtype=torch.int32, device="cuda")
        pr = torch.empty(npoly, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        rec.classify_device(hold["crops"].data_ptr(), npoly, 0, lab.data_ptr(), pr.data_ptr())
        rec.synchronize()
        labels, crops = lab.cpu().numpy(), hold["crops"].cpu().numpy()
    return prob.cpu().numpy(), polys, scores, crops, labels


def bbox(poly):
    xs, ys = [v[0] for v in poly], [v[1] for v in poly]
    return min(xs), min(ys), max(xs), max(ys)


def test_config4_bf16_against_f32_on_8_pages(det_w, rec_w):
    det = capi.Detector(W.pack_blob(det_w), 0)
    rec = capi.Recognizer(W.pack_blob(rec_w), 0)
    frames, boxes = W.synth_text_pages(41, 8, S, S)
    prob32, polys32, scores32, crops32, labels32 = run_pipeline(det, rec, frames, capi.PRECISION_F32)
    prob16, polys16, scores16, crops16, labels16 = run_pipeline(det, rec, frames, capi.PRECISION_BF16)

    # (1) the f32 pipeline against the oracles, stage by stage on the same data
    for i in (0, 5):
        assert np.abs(prob32[i:i + 1] - T.det_forward(det_w, frames[i:i + 1])).max() < 1e-4
    adj = np.ones((8, 2))
    op, os_ = O.get_boxes_and_box_scores(prob32, adj, skip_degenerate=True)
    assert polys32 == op
    assert all(np.allclose(a, b, rtol=0, atol=1e-12) for a, b in zip(scores32, os_))
    assert np.array_equal(crops32, CR.extract_crops(frames, polys32, adj))
    rl, _ = T.rec_classify(T.rec_forward(rec_w, crops32))
    ref_logits = T.rec_forward(rec_w, crops32)
    srt = np.sort(ref_logits, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert (labels32[decided] == rl[decided]).all()
    # every word box of every page is found (the map follows the ink)
    assert [len(p) for p in polys32] == [len(b) for b in boxes]

    # the bf16 pipeline is the same code after the map: its polygons are what the oracle makes of ITS map
    op16, _ = O.get_boxes_and_box_scores(prob16, adj, skip_degenerate=True)
    assert polys16 == op16

    # (2) bf16 against f32: bounded mismatch counts
    d = np.abs(prob16 - prob32)
    flips = int(((prob16 > 0.6) != (prob32 > 0.6)).sum())
    print(f"cfg4 8 pages: max|dprob| {d.max():.3e} mean {d.mean():.3e}, {flips} binarisation flips of {d.size}")
    assert d.max() > 0.0 and d.max() < 0.05 and flips < 2e-4 * d.size
    assert [len(p) for p in polys16] == [len(p) for p in polys32]        # same words found on every page
    n_poly = sum(len(p) for p in polys32)
    moved, identical, k, label_mismatch = 0, 0, 0, 0
    for p32, p16, s32, s16 in zip(polys32, polys16, scores32, scores16):
        for a, b, sa, sb in zip(p32, p16, s32, s16):
            ba, bb = bbox(a), bbox(b)
            assert max(abs(u - v) for u, v in zip(ba, bb)) <= 8            # at most two 4 x 4 cells of the map
            assert abs(sa - sb) < 0.02
            if a == b:
                identical += 1
                assert np.array_equal(crops16[k], crops32[k])               # same polygon -> same crop -> same label
                assert labels16[k] == labels32[k]
            else:
                moved += 1
                label_mismatch += int(labels16[k] != labels32[k])
            k += 1
    print(f"cfg4 8 pages: {n_poly} polygons, {identical} identical, {moved} moved, {label_mismatch} labels differ")
    assert moved <= 0.25 * n_poly and label_mismatch <= 0.1 * n_poly
    det.close()
    rec.close()


def test_config4_128_pages_bf16_properties(det_w, rec_w):
    """128 pages, 4 batches of 32, everything device-resident, bf16 detector.  Size-independent properties: a page's
    polygons / labels do not depend on the batch around it nor on the run; every word box is found exactly once and
    the polygon covers it; labels of identical crops are identical."""
    det = capi.Detector(W.pack_blob(det_w), 0)
    rec = capi.Recognizer(W.pack_blob(rec_w), 0)
    total = 0
    first = None
    for b in range(4):
        frames, boxes = W.synth_text_pages(500 + b, 32, S, S)
        prob, polys, scores, crops, labels = run_pipeline(det, rec, frames, capi.PRECISION_BF16)
        assert np.isfinite(prob).all() and prob.min() >= 0.0 and prob.max() <= 1.0
        k = 0
        for page_polys, page_scores, page_boxes in zip(polys, scores, boxes):
            assert len(page_polys) == len(page_boxes)
            found = set()
            for poly, sc in zip(page_polys, page_scores):
                x0, y0, x1, y1 = bbox(poly)
                hit = [j for j, (bx0, by0, bx1, by1) in enumerate(page_boxes)
                       if x0 <= bx0 + 8 and y0 <= by0 + 8 and x1 >= bx1 - 8 and y1 >= by1 - 8 and x0 >= bx0 - 48 and x1 <= bx1 + 48]
                assert len(hit) == 1 and hit[0] not in found
                found.add(hit[0])
                assert 0.7 <= sc <= 1.0
                k += 1
        assert k == crops.shape[0] == labels.shape[0]
        assert ((labels >= 0) & (labels < 62)).all()
        total += k
        if b == 0:
            first = (frames, polys, scores, labels, crops)
    assert total > 128 * 10
    # batch independence and idempotence on the first batch: pages 3 and 17 alone, and the whole batch again
    frames, polys, scores, labels, crops = first
    again = run_pipeline(det, rec, frames, capi.PRECISION_BF16)
    assert again[1] == polys and again[2] == scores and np.array_equal(again[4], labels)
    offs = np.cumsum([0] + [len(p) for p in polys])
    for i in (3, 17):
        one = run_pipeline(det, rec, frames[i:i + 1], capi.PRECISION_BF16)
        assert one[1][0] == polys[i] and one[2][0] == scores[i]
        assert np.array_equal(one[4], labels[offs[i]:offs[i + 1]])
    det.close()
    rec.close()


def test_pipelined_detection_equals_batch_by_batch(det_w):
    """ocr_det_detect_pipelined overlaps the post-processing of batch k with the forward of batch k + 1 (second stream,
    host thread pool); what it returns must be exactly forward + get_boxes_and_box_scores of each batch, in order."""
    import torch
    det = capi.Detector(W.pack_blob(det_w), 0)
    params = capi.default_params(skip_degenerate=True)
    batches = [W.synth_text_pages(900 + b, 3 + b, S, S)[0] for b in range(4)]     # ragged batch sizes 3, 4, 5, 6
    want = []
    for fr in batches:
        prob = det.forward_host(fr)
        want.append(det.postprocess(prob, fr.shape[0], S, S, np.ones((fr.shape[0], 2)), capi.MEM_HOST, params))
    xs = [torch.from_numpy(fr).cuda() for fr in batches]
    probs = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    got = []
    for x, pr in zip(xs, probs):
        r = det.detect_pipelined(x.data_ptr(), x.shape[0], S, S, pr.data_ptr(), np.ones((x.shape[0], 2)), params)
        got.append(r)
    got.append(det.detect_pipelined(0, 0, 0, 0, 0))            # flush: the last batch
    assert got[0] is None and det.detect_pipelined(0, 0, 0, 0, 0) is None   # nothing pending any more
    assert got[1:] == want
    assert all(sum(len(p) for p in polys) > 0 for polys, _ in want)
    det.close()

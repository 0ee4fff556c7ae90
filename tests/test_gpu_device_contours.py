"""The device contour tracer (contours.hip) against the host tracer it restates (postproc_geom.cpp, itself pinned to
imageproc's find_contours through the reference's known answers): the SAME contours, point for point, in the same order - start
pixels and order decide what Douglas-Peucker keeps - and the same polygons out of ocr_det_postprocess whichever tracer ran."""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from tests import fixtures as FX
from oracle import postproc_oracle as O

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(11)
    out = []
    for h, w, p in ((32, 32, 0.5), (64, 96, 0.3), (96, 64, 0.7), (160, 160, 0.5), (33, 64, 0.45), (7, 32, 0.6), (200, 224, 0.1)):
        out.append((f"noise {h}x{w} p={p}", (rng.random((h, w)) < p).astype(np.uint8)))
    m = rng.random((256, 256))
    for _ in range(40):
        m = (m + np.roll(m, 1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 0) + np.roll(m, -1, 1)) / 5
    blobs = (m > np.median(m)).astype(np.uint8)
    out.append(("smooth blobs", blobs.copy()))
    blobs[:, 0] = 0
    out.append(("smooth blobs, column 0 clear", blobs))
    ring = np.zeros((96, 128), np.uint8)
    ring[10:40, 10:50] = 1; ring[11:39, 11:49] = 0          # one pixel thick: outer and hole border share every pixel
    ring[50:90, 20:100] = 1; ring[55:85, 25:95] = 0; ring[60:80, 30:90] = 1; ring[65:75, 35:85] = 0   # nested
    ring[5, 60:120] = 1; ring[5:45, 120] = 1                # thin lines
    out.append(("rings and lines", ring))
    out.append(("full", np.ones((64, 64), np.uint8)))
    out.append(("empty", np.zeros((64, 64), np.uint8)))
    edge = np.zeros((64, 96), np.uint8)
    edge[0, :] = 1; edge[:, 0] = 1; edge[-1, 3:40] = 1; edge[10:30, -1] = 1; edge[20:25, 90:] = 1; edge[40, 40] = 1; edge[63, 95] = 1
    out.append(("touching every border", edge))
    out.append(("checkerboard", ((np.add.outer(np.arange(64), np.arange(64))) % 2).astype(np.uint8)))
    out.append(("stripes", (np.arange(128)[None, :] % 3 == 0).astype(np.uint8).repeat(40, 0)))
    out.append(("diagonals", (np.add.outer(np.arange(96), np.arange(96)) % 7 < 2).astype(np.uint8)))
    out.append(("text-like 640", (FX.text_like_maps(1, 640, 3)[0, 0] > 0.6).astype(np.uint8)))
    out.append(("dense 640", (FX.dense_text_maps(1, 640, 4)[0, 0] > 0.6).astype(np.uint8)))
    out.append(("text-like 640 b", (FX.text_like_maps(1, 640, 13)[0, 0] > 0.6).astype(np.uint8)))
    out.append(("dense 320", (FX.dense_text_maps(1, 320, 14)[0, 0] > 0.6).astype(np.uint8)))
    # the scan streams the bit image through a 64-row band buffer: heights around its edges, the widest and the tallest map the
    # parallel form takes, a non-square page
    for h, w in ((63, 64), (65, 64), (127, 96), (129, 160), (64, 2048), (1024, 32), (320, 480)):
        m = rng.random((h, w))
        for _ in range(6):
            m = (m + np.roll(m, 1, 0) + np.roll(m, 1, 1) + np.roll(m, -1, 0) + np.roll(m, -1, 1)) / 5
        bm = (m > np.quantile(m, 0.7)).astype(np.uint8)
        bm[:, 0] = 0
        out.append((f"blobs {h}x{w}", bm))
    return out


CASES = _cases()


@pytest.mark.parametrize("sequential", [False, True], ids=["parallel", "one-wave"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_contours_equal_host_contours(case, sequential):
    name, bm = case
    want = capi.host_contours(bm)
    got, status = capi.device_contours(bm, sequential=sequential)
    if status == 3:
        # the parallel form met a start outside its list of plausible starts and gave the image up (the engine sends it to the host
        # tracer): only where the implementation's x > 0 rule moves a component's outer start - foreground in column 0
        assert not sequential and bm[:, 0].any(), name
        return
    if status == 1 and not sequential:
        # ... or its speculative walks outgrew their pool: noise, where thousands of plausible starts sit on one giant border
        assert name.startswith("noise") or name.startswith("blobs ") or name in ("checkerboard", "diagonals", "smooth blobs"), name
        return
    assert status == 0
    assert len(got) == len(want), (len(got), len(want))
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, f"contour {k} differs"
    # ... and the ORACLE's contours (oracle/postproc_oracle.py: the restatement of imageproc's find_contours that the reference's
    # known answers pin), not only the product's host tracer; the pure-Python walk is kept to the maps it finishes in seconds
    if bm.size <= 160 * 160 or name in ("rings and lines", "touching every border", "text-like 640", "dense 320"):
        oracle = [[(int(x), int(y)) for x, y in c] for c in O.find_contours(bm * 255)]
        assert len(got) == len(oracle), (len(got), len(oracle))
        for k, (a, b) in enumerate(zip(got, oracle)):
            assert [tuple(q) for q in a] == b, f"contour {k} differs from the oracle's"


def test_parallel_form_takes_the_usual_maps():
    """... and does not give up (status 3) on maps without foreground in column 0: noise, blobs, rings, text-like and dense maps."""
    taken = 0
    for name, bm in CASES:
        if bm[:, 0].any() or bm.shape[0] > 1024 or name.startswith("noise") or name.startswith("blobs ") or name in ("checkerboard", "diagonals"):
            continue
        _, status = capi.device_contours(bm)
        assert status == 0, name
        taken += 1
    assert taken >= 5


def test_device_contours_report_overflow_and_guard():
    bm = (np.random.default_rng(3).random((64, 64)) < 0.4).astype(np.uint8)
    bm[:, 0] = 0   # (foreground in column 0 makes the parallel form give up with status 3 before anything overflows)
    want = capi.host_contours(bm)
    assert len(want) > 8
    for seq in (False, True):
        _, status = capi.device_contours(bm, max_pts=1 << 16, max_polys=8, sequential=seq)       # more contours than the buffer holds
        assert status == 1
        _, status = capi.device_contours(bm, max_pts=16, max_polys=1 << 12, sequential=seq)      # more points
        assert status == 1
    _, status = capi.device_contours(np.zeros((800, 800), np.uint8))          # the reference's frame size fits (one bit plane + a band of rows in LDS)
    assert status == 0
    with pytest.raises(capi.OcrError):
        capi.device_contours(np.zeros((1024, 2048), np.uint8))                # a bit plane of 1024 x 2048 does not fit a CU's LDS


def _post(det, maps, adj):
    p = capi.default_params(skip_degenerate=True)
    return det.postprocess(maps, maps.shape[0], maps.shape[2], maps.shape[3], adj, capi.MEM_HOST, p)


def test_postprocess_is_the_same_with_either_tracer():
    """ocr_det_postprocess with device_contours=1 (parallel form) / 2 (one wave per image) / 0 - with the contours the whole chain
    (Douglas-Peucker, job list, box scores, unclip: candidates.hip, unclip.hip) stays on the device unless device_polygons=0 -: identical polygon lists and
    scores on text-like and dense maps, on noise whose thousands of contours overflow the device buffers (those images fall back to the
    host tracer inside the call), on a mixed batch, and at the reference's 800 x 800."""
    blob = W.pack_blob(W.make_det_weights(0))
    host = capi.Detector(blob, 0, options="device_contours=0")
    dev = capi.Detector(blob, 0, options="device_contours=1")
    dev2 = capi.Detector(blob, 0, options="device_contours=2;post_threads=2")     # the one-wave-per-image form
    dev3 = capi.Detector(blob, 0, options="device_contours=1;device_polygons=0")  # contours back to the host: Douglas-Peucker on the pool
    dev4 = capi.Detector(blob, 0, options="device_contours=1;device_unclip=0")    # (no device unclip -> no device chain either)
    rng = np.random.default_rng(5)
    noise = (rng.random((2, 1, 640, 640)) * 0.9).astype(np.float32)         # ~ 40 k contours per image: overflow -> host fallback
    smooth = rng.random((2, 1, 320, 320)).astype(np.float32)
    for _ in range(5):
        smooth = (smooth + np.roll(smooth, 1, 2) + np.roll(smooth, 1, 3) + np.roll(smooth, -1, 2) + np.roll(smooth, -1, 3)) / 5
    smooth = ((smooth - smooth.min()) / (smooth.max() - smooth.min())).astype(np.float32)
    mixed = np.concatenate([FX.dense_text_maps(2, 640, 7), noise[:1], FX.text_like_maps(1, 640, 8)])
    for name, maps in (("text", FX.text_like_maps(4, 640, 1)), ("dense", FX.dense_text_maps(4, 640, 2)), ("noise", noise), ("smooth", smooth),
                       ("mixed", mixed), ("800", FX.text_like_maps(1, 800, 9))):
        adj = np.ones((maps.shape[0], 2)) * np.array([1.25, 0.8])
        want = _post(host, maps, adj)
        for d in (dev, dev2, dev3, dev4):
            got = _post(d, maps, adj)
            assert got[0] == want[0], name
            assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got[1], want[1])), name
    for d in (host, dev, dev2, dev3, dev4):
        d.close()
    with pytest.raises(capi.OcrError):
        capi.Detector(blob, 0, options="device_contours=3")


def test_pipelined_calls_with_pretraced_contours_return_the_same_polygons():
    """device_contours in the pipelined calls: the contours of a batch are requested when the batch is queued (own stream, behind its
    forward) and only read by the call that brings its polygons back.  Same polygons and scores as the host tracer, batch after
    batch: device frames and host frames, batches of changing size (the scratch slot and the staging slots grow while a batch is
    pending), a crop extraction between two calls, a flush, and an ordinary ocr_det_postprocess in between."""
    import torch
    det_w = W.make_det_weights_text()
    blob = W.pack_blob(det_w)
    params = capi.default_params(skip_degenerate=True)
    sizes = [3, 5, 2, 8, 8, 1]
    pages = [W.synth_text_pages(300 + i, n, 640, 640, dense=bool(i & 1))[0] for i, n in enumerate(sizes)]
    results = {}
    for dc in (0, 1, 2, 3):
        det = capi.Detector(blob, 0, options=f"device_contours={dc};post_threads=2" if dc < 3 else "device_contours=1;device_polygons=0;post_threads=2")
        got = []
        # device frames
        dev = [torch.from_numpy(p).cuda() for p in pages]
        prob = [torch.empty_like(d) for d in dev]
        crops = torch.empty((4096, 784), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for i, d in enumerate(dev):
            n = d.shape[0]
            blk = det.detect_pipelined_block(d.data_ptr(), n, 640, 640, prob[i].data_ptr(), np.ones((n, 2)), params)
            if blk is not None:
                pn = dev[i - 1].shape[0]
                det.extract_crops_block(blk, dev[i - 1].data_ptr(), pn, 640, 640, np.ones((pn, 2)), crops.data_ptr())   # between two calls
                got.append(capi.polygons_to_python(blk))
                det.free_block(blk)
            if i == 2:   # an ordinary post-processing call while a pretraced batch is pending
                m = np.ascontiguousarray(prob[0].cpu().numpy())
                got.append(det.postprocess(m, m.shape[0], 640, 640, np.ones((m.shape[0], 2)), capi.MEM_HOST, params))
        got.append(det.detect_pipelined(0, 0, 0, 0, 0))
        # host frames (u8, then f32: the staging slots grow)
        for i, p in enumerate(pages[:4]):
            x = np.clip(np.rint(p), 0, 255).astype(np.uint8) if i < 2 else np.ascontiguousarray(np.rint(p).astype(np.float32))
            r = det.detect_pipelined_host(x, adjust_values=np.ones((x.shape[0], 2)), params=params)
            if r is not None:
                got.append(r)
        got.append(det.detect_pipelined_host(None))
        det.close()
        results[dc] = got
    assert len(results[0]) == len(results[1]) == len(results[2]) == len(results[3]) and len(results[0]) >= 10
    for dc in (1, 2, 3):
        for k, (a, b) in enumerate(zip(results[0], results[dc])):
            assert a[0] == b[0], (dc, k)
            assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a[1], b[1])), (dc, k)
    assert sum(len(p) for r in results[0] for p in r[0]) > 100   # there was something to compare


def test_device_chain_overflow_takes_the_host_path():
    """More candidates than the device job list holds (1 024 per image of the batch): the chain reports the overflow with ZERO jobs - so that
    box scores and unclip, which walk the list by that count, touch nothing - and the host path takes the batch: same polygons."""
    m = np.zeros((1, 1, 640, 640), np.float32)
    k = 0
    for gy in range(8, 632, 16):
        for gx in range(8, 632, 20):
            if k < 1200:
                m[0, 0, gy:gy + 7, gx:gx + 9] = 0.9
                k += 1
    blob = W.pack_blob(W.make_det_weights(0))
    host = capi.Detector(blob, 0, options="device_contours=0;device_unclip=0")
    dev = capi.Detector(blob, 0, options="device_contours=1")
    adj = np.ones((1, 2))
    want = _post(host, m, adj)
    got = _post(dev, m, adj)
    assert len(want[0][0]) == 1200
    assert got[0] == want[0] and got[1] == want[1]
    # ... and a batch of two such maps with room for both (2 048 jobs) stays on the device and agrees as well
    m2 = np.concatenate([m, m])
    m2[1, 0, 300:, :] = 0.0
    want2 = _post(host, m2, np.ones((2, 2)))
    got2 = _post(dev, m2, np.ones((2, 2)))
    assert got2[0] == want2[0] and got2[1] == want2[1]
    host.close()
    dev.close()

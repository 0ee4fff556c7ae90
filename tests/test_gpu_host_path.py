"""The host-memory entry points - what a drop-in for the reference's call sites binds (CPU tensors in, polygon lists out,
/root/reference/src/text_detection/mod.rs:46-67) - against the device-pointer forms they are built from:
  ocr_det_forward_u8            the u8 image itself (image_ops.rs:350-364), bit for bit the f32 entry on (float)x
  ocr_det_forward (host)        pieces of the batch pipelined over copy-in / forward / copy-out streams
  ocr_det_detect_pipelined_host frames from host memory (pageable or pinned, f32 or u8), maps optional, polygons identical
"""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    d = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
    yield d
    d.close()


def _u8_frames(seed, n, h, w):
    return np.random.default_rng(seed).integers(0, 256, (n, 1, h, w), dtype=np.uint8)


@pytest.mark.parametrize("options", [None, "mfma=f32", "precision=bf16"])
def test_u8_entry_is_the_f32_entry_bit_for_bit(options):
    import torch
    d = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options=options)
    try:
        xb = _u8_frames(3, 3, 96, 160)
        xf = xb.astype(np.float32)
        want = d.forward_host(xf)
        assert np.array_equal(d.forward_host_u8(xb), want)                      # host memory
        xd = torch.from_numpy(xb).cuda()
        pd = torch.empty((3, 1, 96, 160), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        d.forward_u8_device(xd.data_ptr(), 3, 96, 160, pd.data_ptr())           # device memory
        assert np.array_equal(pd.cpu().numpy(), want)
    finally:
        d.close()


def test_host_forward_in_pieces_equals_one_device_forward(det):
    """20 frames run as pieces of 8, 8 and 4 through the staging slots; frames are independent (eval-mode batch norm), so
    the maps are those of one forward over device-resident frames."""
    import torch
    x = W.synth_image_batch(5, 20, 64, 96)
    xd = torch.from_numpy(x).cuda()
    pd = torch.empty_like(xd)
    torch.cuda.synchronize()
    det.forward_device(xd.data_ptr(), 20, 64, 96, pd.data_ptr())
    det.synchronize()
    want = pd.cpu().numpy()
    assert np.array_equal(det.forward_host(x), want)
    assert np.array_equal(det.forward_host(x[:3]), want[:3])     # one short piece, staging reused
    pin_in, pin_out = capi.HostBuffer(x.shape, np.float32), capi.HostBuffer(x.shape, np.float32)
    pin_in.array[...] = x
    capi.check(capi.lib().ocr_det_forward(det._h, pin_in.array.ctypes.data, 20, 64, 96, pin_out.array.ctypes.data, capi.MEM_HOST))
    assert np.array_equal(pin_out.array, want)                   # pinned memory: asynchronous copies
    pin_in.close()
    pin_out.close()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
@pytest.mark.parametrize("u8", [False, True], ids=["f32", "u8"])
def test_pipelined_detection_from_host_memory(pinned, u8):
    """Polygon lists (and, when asked for, maps) of ocr_det_detect_pipelined_host are exactly forward +
    get_boxes_and_box_scores per batch, for ragged batch sizes, both element kinds and both kinds of host memory."""
    S = 320
    d = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0)
    params = capi.default_params(skip_degenerate=True)
    batches = [W.synth_text_pages(700 + b, 2 + b, S, S)[0] for b in range(5)]     # 2, 3, 4, 5, 6 pages: slots alternate
    if u8:
        batches = [np.clip(np.rint(b), 0, 255).astype(np.uint8) for b in batches]
    want, want_maps = [], []
    for fr in batches:
        prob = d.forward_host(fr.astype(np.float32))
        want_maps.append(prob)
        want.append(d.postprocess(prob, fr.shape[0], S, S, np.ones((fr.shape[0], 2)), capi.MEM_HOST, params))
    bufs, maps, hold = [], [], []
    for fr in batches:
        if pinned:
            hb = capi.HostBuffer(fr.shape, fr.dtype)
            hb.array[...] = fr
            hm = capi.HostBuffer(fr.shape, np.float32)
            hold += [hb, hm]
            bufs.append(hb.array)
            maps.append(hm.array)
        else:
            bufs.append(fr)
            maps.append(np.zeros(fr.shape, np.float32))
    got = []
    for i, fr in enumerate(bufs):
        got.append(d.detect_pipelined_host(fr, adjust_values=np.ones((fr.shape[0], 2)), prob_out=maps[i] if i % 2 == 0 else None,
                                           params=params))
    got.append(d.detect_pipelined_host(None))
    assert got[0] is None and d.detect_pipelined_host(None) is None
    assert got[1:] == want
    assert all(sum(len(p) for p in polys) > 0 for polys, _ in want)
    for i in range(0, len(bufs), 2):
        assert np.array_equal(maps[i], want_maps[i])
    d.close()
    for hb in hold:
        hb.close()


def test_pipelined_host_staging_growth_and_interleaved_host_forward():
    """A FRESH handle (staging sized by the first pipelined batch), batches that grow from call to call, a switch from u8 to f32
    frames without a flush (4 x the input bytes) and a blocking host forward between two pipelined calls: the pending batch's
    map lives in a staging slot across calls, and none of these may free or overwrite it (round-3 advisor finding)."""
    S = 320
    params = capi.default_params(skip_degenerate=True)
    ref = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0)
    d = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0)
    pages = [W.synth_text_pages(900 + b, 2 + b, S, S)[0] for b in range(5)]                     # 2, 3, 4, 5, 6 pages: every call grows
    batches = [np.clip(np.rint(b), 0, 255).astype(np.uint8) if i < 3 else np.rint(b).astype(np.float32) for i, b in enumerate(pages)]
    want, want_maps = [], []
    for fr in batches:
        prob = ref.forward_host(fr.astype(np.float32))
        want_maps.append(prob)
        want.append(ref.postprocess(prob, fr.shape[0], S, S, np.ones((fr.shape[0], 2)), capi.MEM_HOST, params))
    maps = [np.zeros(fr.shape, np.float32) for fr in batches]
    other = W.synth_image_batch(3, 9, 64, 96)
    other_want = ref.forward_host(other)
    got = []
    for i, fr in enumerate(batches):
        got.append(d.detect_pipelined_host(fr, adjust_values=np.ones((fr.shape[0], 2)), prob_out=maps[i], params=params))
        if i == 1:   # a blocking host forward of another shape while batch 1 is pending
            assert np.array_equal(d.forward_host(other), other_want)
    got.append(d.detect_pipelined_host(None))
    assert got[0] is None
    assert got[1:] == want
    assert all(sum(len(p) for p in polys) > 0 for polys, _ in want)
    for m, wm in zip(maps, want_maps):
        assert np.array_equal(m, wm)
    d.close()
    ref.close()


def test_post_threads_option(det):
    """post_threads sizes the host pool of the post-processing stages; results do not depend on it."""
    S = 320
    fr = W.synth_text_pages(41, 6, S, S)[0]
    params = capi.default_params(skip_degenerate=True)
    outs = []
    for opt in ("post_threads=1", "post_threads=3", None):
        d = capi.Detector(W.pack_blob(W.make_det_weights_text()), 0, options=opt)
        prob = d.forward_host(fr)
        outs.append(d.postprocess(prob, 6, S, S, np.ones((6, 2)), capi.MEM_HOST, params))
        d.close()
    assert outs[0] == outs[1] == outs[2] and sum(len(p) for p in outs[0][0]) > 0
    with pytest.raises(capi.OcrError):
        capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options="post_threads=-2")

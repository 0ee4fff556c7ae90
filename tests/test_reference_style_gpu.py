"""The reference's own post-processing tests (metrics.rs:400-646), restated against
the drop-in API of ocr-rs_amd/text_detection.py + char_recognition.py on the GPU."""
import os

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import char_recognition as cr
from ocr_rs_amd import text_detection as td
from ocr_rs_amd import weights as W
from tests import kat_postproc as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    n = td.resnet18(W.pack_blob(W.make_det_weights(0)), 0)
    yield n
    n.close()


def _pred_from_png(golden_dir):
    # metrics.rs:512-517: open(..).to_luma(); pred = convert_image_to_tensor(img) / 255
    img = np.array(Image.open(os.path.join(golden_dir, "gt_shrinked_img55.png")).convert("L"))
    return (img.astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)


def test_get_polygons_from_bitmap_test_without_adjustments(net, golden_dir):   # metrics.rs:510-580
    res = td.get_boxes_and_box_scores(net, _pred_from_png(golden_dir), np.array([[1.0, 1.0]]))
    assert res.polygons[0] == K.IMG55_POLYS_ADJ1
    assert res.scores[0] == K.IMG55_SCORES


def test_get_polygons_from_bitmap_test_with_2x_adjustments(net, golden_dir):   # metrics.rs:582-646
    res = td.get_boxes_and_box_scores(net, _pred_from_png(golden_dir), np.array([[2.0, 2.0]]))
    assert res.polygons[0] == K.IMG55_POLYS_ADJ2
    assert res.scores[0] == K.IMG55_SCORES


def test_binarize_test(net):                                                     # metrics.rs:486-508
    # binarize is fused into the post-processing entry; observe it through the contours:
    # the KAT map thresholded at 0.57 leaves exactly the KAT's 0/1 pattern.
    import torch
    vals = torch.tensor(K.BINARIZE_IN, dtype=torch.float32).reshape(1, 1, 5, 5)
    big = torch.zeros(1, 1, 32, 32)
    big[..., :5, :5] = vals
    x = big.cuda()
    bm = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    # the fused kernel thresholds the detector's own output; the standalone binarize is the
    # one post-processing uses: reach it with a params override
    p = capi.default_params(skip_degenerate=True)
    p.thresh = K.BINARIZE_THRESH
    expected = (big > np.float32(K.BINARIZE_THRESH)).to(torch.uint8)
    assert expected[0, 0, :5, :5].reshape(-1).tolist() == K.BINARIZE_OUT
    polys, _ = net.handle.postprocess(big.numpy(), 1, 32, 32, np.array([[1.0, 1.0]]), capi.MEM_HOST, p)
    from oracle import postproc_oracle as O
    assert polys == O.get_boxes_and_box_scores(big.numpy(), np.array([[1.0, 1.0]]), thresh=K.BINARIZE_THRESH,
                                               skip_degenerate=True)[0]
    del x, bm


def test_forward_t_device_tensor_and_train_flag(net):
    import torch
    from oracle import torch_ref as T
    x = W.synth_image_batch(3, 1, 64, 64)
    ref = T.det_forward(W.make_det_weights(0), x)
    # the contract of a torch op: no manual synchronisation around the call, on the default stream ...
    y = net.forward_t(torch.from_numpy(x).cuda())
    assert np.abs(y.cpu().numpy() - ref).max() < 1e-4
    # ... and on a side stream, where producer (the copy), forward and consumer are ordered by the stream alone
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xs = torch.from_numpy(x).pin_memory().cuda(non_blocking=True)
        y2 = net.forward_t(xs)
        host = y2.to("cpu", non_blocking=False)
    assert np.abs(host.numpy() - ref).max() < 1e-4
    with pytest.raises(capi.OcrError):
        net.forward_t(x, train=True)


def test_char_recognition_run_prediction_tail():
    from oracle import torch_ref as T
    rw = W.make_rec_weights(0)
    n = cr.Net(W.pack_blob(rw), 0)
    crops = W.synth_crops(2, 16)
    out = n.predict(crops)
    lab, pr = T.rec_classify(T.rec_forward(rw, crops))
    assert [c for c, _ in out] == [cr.VALUES[i] for i in lab]
    assert np.allclose([p for _, p in out], pr, atol=1e-5)
    with pytest.raises(capi.OcrError):
        n.forward_t(np.zeros((1, 32 * 128), np.float32)[:, :100])     # not a multiple of 784 (view would fail)
    n.close()

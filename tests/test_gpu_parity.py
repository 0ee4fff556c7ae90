"""Parity of the HIP path (through the C ABI) with the oracle, on a real MI355X.
Bars (BASELINE.json north_star): fp32 probability maps within 1e-4, polygon vertex
indices and labels bit-exact."""
import os

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import cnn_oracle as CO
from oracle import postproc_oracle as O
from oracle import torch_ref as T
from tests import kat_postproc as K

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def det_w():
    return W.make_det_weights(0)


@pytest.fixture(scope="module")
def det(det_w):
    d = capi.Detector(W.pack_blob(det_w), 0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def rec_w():
    return W.make_rec_weights(0)


@pytest.fixture(scope="module")
def rec(rec_w):
    r = capi.Recognizer(W.pack_blob(rec_w), 0)
    yield r
    r.close()


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_det_stagewise_parity(det, det_w):
    """Every fused stage against the oracle's activations (relative to the stage's scale)."""
    x = W.synth_image_batch(7, 2, 64, 96)
    st = {}
    ref = T.det_forward(det_w, x, st)
    prob = det.forward_host(x)
    n, h, w = 2, 64, 96
    got = {
        "stem": det.debug_stage(0, (n, h // 4, w // 4, 64)),
        "layer1": det.debug_stage(1, (n, h // 4, w // 4, 64)),
        "layer2": det.debug_stage(2, (n, h // 8, w // 8, 128)),
        "layer3": det.debug_stage(3, (n, h // 16, w // 16, 256)),
        "layer4": det.debug_stage(4, (n, h // 32, w // 32, 512)),
        "bin1": det.debug_stage(13, (n, h // 4, w // 4, 64)),
    }
    for k, v in got.items():
        assert _rel(v, st[k]) < 2e-5, (k, _rel(v, st[k]))
    assert np.abs(prob - ref).max() < TOL


def test_det_matches_committed_golden(det, golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    x = W.synth_image_batch(int(g["det_input_seed"]), 2, 64, 96)
    assert np.abs(det.forward_host(x) - g["det_prob"]).max() < TOL


@pytest.mark.parametrize("n,h,w,seed", [(1, 32, 32, 3), (3, 96, 160, 4), (1, 160, 96, 5), (2, 256, 256, 6)])
def test_det_forward_parity_shapes(det, det_w, n, h, w, seed):
    x = W.synth_image_batch(seed, n, h, w)
    prob = det.forward_host(x)
    assert prob.shape == (n, 1, h, w)
    assert np.abs(prob - T.det_forward(det_w, x)).max() < TOL


def test_det_forward_parity_against_plain_c_oracle(det, det_w):
    x = W.synth_image_batch(9, 1, 64, 64)
    assert np.abs(det.forward_host(x) - CO.det_forward(det_w, x)).max() < TOL


def test_det_other_weights_seed():
    w2 = W.make_det_weights(5)
    d = capi.Detector(W.pack_blob(w2), 0)
    x = W.synth_image_batch(2, 1, 64, 128)
    assert np.abs(d.forward_host(x) - T.det_forward(w2, x)).max() < TOL
    d.close()


def test_det_two_kernel_head_matches_fused_head(det_w):
    """Option tail_unfused=1 keeps bin_conv_tr1 / bin_conv_tr2 as two launches; both heads must agree."""
    x = W.synth_image_batch(12, 2, 64, 96)
    d2 = capi.Detector(W.pack_blob(det_w), 0, options="tail_unfused=1")
    d1 = capi.Detector(W.pack_blob(det_w), 0)
    a, b = d1.forward_host(x), d2.forward_host(x)
    ref = T.det_forward(det_w, x)
    assert np.abs(a - ref).max() < TOL and np.abs(b - ref).max() < TOL
    assert np.abs(a - b).max() < 1e-5
    d1.close()
    d2.close()


def test_det_preprocessed_img55_plumbing(det, det_w, golden_dir):
    """BASELINE config 0: the reference's own 800x800 fixture through detect -> polygons."""
    img = np.array(Image.open(os.path.join(golden_dir, "preprocessed_img55.png")).convert("L"))
    x = img.astype(np.float32).reshape(1, 1, 800, 800)
    prob = det.forward_host(x)
    ref = T.det_forward(det_w, x)
    assert np.abs(prob - ref).max() < TOL
    adj = np.array([[800 / 300, 533 / 200]])          # image_ops.rs:910-916
    # a random-weight map is noise: it holds zero-area contours on which the reference
    # itself aborts (metrics.rs:103 unwrap); both sides must report exactly that ...
    with pytest.raises(O.DegeneratePolygon):
        O.get_boxes_and_box_scores(prob, adj)
    with pytest.raises(capi.OcrError) as e:
        det.postprocess(prob, 1, 800, 800, adj)
    assert e.value.code == 6
    # ... and agree on everything else when such candidates are dropped
    polys, scores = det.postprocess(prob, 1, 800, 800, adj, params=capi.default_params(skip_degenerate=True))
    opolys, oscores = O.get_boxes_and_box_scores(prob, adj, skip_degenerate=True)
    assert polys == opolys and len(polys[0]) > 0
    assert np.allclose(scores[0], oscores[0], rtol=0, atol=1e-12)


def test_det_rejects_bad_shapes(det):
    with pytest.raises(capi.OcrError) as e:
        det.forward_host(np.zeros((1, 1, 40, 64), np.float32))
    assert e.value.code == 1


def test_fused_binarize_matches_threshold(det):
    import torch
    x = torch.from_numpy(W.synth_image_batch(8, 2, 64, 64)).cuda()
    prob = torch.empty_like(x)
    bm = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    det.forward_device(x.data_ptr(), 2, 64, 64, prob.data_ptr(), bm.data_ptr(), 0.6)
    det.synchronize()
    assert torch.equal(bm, (prob > 0.6).to(torch.uint8))          # metrics.rs:129-131, f32 compare
    assert 0 < int(bm.sum()) < bm.numel()


# ---------------------------------------------------------------- post-processing
def _img(golden_dir, name):
    return np.array(Image.open(os.path.join(golden_dir, name)).convert("L"))


@pytest.mark.parametrize("adj,expected", [((1.0, 1.0), K.IMG55_POLYS_ADJ1), ((2.0, 2.0), K.IMG55_POLYS_ADJ2)])
def test_postprocess_reference_kat(det, golden_dir, adj, expected):
    """metrics.rs:510-646 through the C ABI (GPU binarize + GPU box score + host geometry)."""
    img = _img(golden_dir, "gt_shrinked_img55.png")
    pred = (img.astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)
    polys, scores = det.postprocess(pred, 1, 800, 800, np.array([adj]))
    assert polys[0] == expected
    assert scores[0] == K.IMG55_SCORES


def test_box_score_kernel_reference_kats(det):
    """metrics.rs:426-484: box_score_fast on the 5x5 map, straight on the GPU kernel."""
    pred = np.array(K.BOX_SCORE_MAP, dtype=np.float32).reshape(5, 5)
    sums, counts = det.debug_box_scores(pred, [pts for pts, _ in K.BOX_SCORE_CASES])
    assert (sums / counts).tolist() == [exp for _, exp in K.BOX_SCORE_CASES]
    assert counts.tolist() == [25.0, 16.0, 12.0]


def test_box_score_kernel_matches_oracle_masks(det):
    """Mask pixel counts and f64 sums of thin / degenerate / concave polygons."""
    rng = np.random.RandomState(1)
    pred = rng.rand(96, 96).astype(np.float32)
    polys = [[(76, 36), (74, 38), (74, 39), (74, 38)],                       # zero-area line
             [(10, 10), (40, 12), (38, 30), (25, 18), (12, 33)],              # concave
             [(50, 50), (90, 50), (90, 90), (50, 90)],                        # square
             [(5, 60), (30, 61), (29, 80), (6, 95), (5, 70), (20, 72)],       # self-touching
             [(60, 5), (61, 30), (62, 5), (63, 30)]]                          # zig-zag sliver
    sums, counts = det.debug_box_scores(pred, polys)
    for k, p in enumerate(polys):
        assert counts[k] == O.polygon_mask_count(p, pred.shape), (k, counts[k])
        assert abs(sums[k] / counts[k] - O.box_score_fast(pred, p)) < 1e-13


def test_box_score_large_polygon_bands(det):
    """A bounding box larger than one 32 KiB mask band (row-banded path)."""
    rng = np.random.RandomState(2)
    pred = rng.rand(800, 800).astype(np.float32)
    poly = [(20, 30), (770, 15), (790, 700), (400, 780), (30, 760), (200, 400)]
    sums, counts = det.debug_box_scores(pred, [poly])
    assert counts[0] == O.polygon_mask_count(poly, pred.shape)
    assert abs(sums[0] / counts[0] - O.box_score_fast(pred, poly)) < 1e-13


def test_postprocess_small_map_matches_oracle(det):
    pred = np.zeros((1, 1, 32, 32), np.float32)
    pred[0, 0, 4:20, 6:26] = 0.9
    polys, scores = det.postprocess(pred, 1, 32, 32, np.array([[1.0, 1.0]]))
    opolys, oscores = O.get_boxes_and_box_scores(pred, np.array([[1.0, 1.0]]))
    assert polys == opolys and len(polys[0]) == 1
    assert scores[0] == oscores[0]


def test_postprocess_batch_matches_oracle(det, golden_dir):
    names = ["gt_shrinked_img55.png", "gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]
    rng = np.random.RandomState(0)
    maps = []
    for nm in names:
        m = _img(golden_dir, nm).astype(np.float32) / 255.0
        # real-valued probabilities: inside ~U(0.55,1), outside ~U(0,0.45); some blobs fail box_thresh
        m = np.where(m > 0.5, 0.55 + 0.45 * rng.rand(800, 800), 0.45 * rng.rand(800, 800)).astype(np.float32)
        maps.append(m)
    pred = np.stack(maps)[:, None]
    adj = np.array([[1.0, 1.0], [2.6666666666666665, 2.665], [0.5, 0.75], [1.3, 1.0]])
    polys, scores = det.postprocess(pred, 4, 800, 800, adj)
    opolys, oscores = O.get_boxes_and_box_scores(pred, adj)
    assert polys == opolys
    assert sum(len(p) for p in polys) >= 8
    for a, b in zip(scores, oscores):
        assert np.allclose(a, b, rtol=0, atol=1e-12)


def test_postprocess_noise_blobs_match_oracle(det):
    rng = np.random.RandomState(3)
    f = rng.rand(2, 160, 224)
    for _ in range(4):
        f = (f + np.roll(f, 1, 1) + np.roll(f, -1, 1) + np.roll(f, 1, 2) + np.roll(f, -1, 2)) / 5
    f = (f - f.min()) / (f.max() - f.min())
    pred = np.clip(0.6 + (f - np.median(f)) * 6, 0, 1).astype(np.float32)[:, None]
    adj = np.array([[1.0, 1.0], [1.4, 0.8]])
    polys, scores = det.postprocess(pred, 2, 160, 224, adj, params=capi.default_params(skip_degenerate=True))
    opolys, oscores = O.get_boxes_and_box_scores(pred, adj, skip_degenerate=True)
    assert polys == opolys and sum(len(p) for p in polys) > 2
    for a, b in zip(scores, oscores):   # NaN = empty mask on a non-square map (the reference's x/y clamp quirk)
        assert np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True)


def test_postprocess_empty_map(det):
    polys, scores = det.postprocess(np.zeros((2, 1, 64, 64), np.float32), 2, 64, 64, np.ones((2, 2)))
    assert polys == [[], []] and scores == [[], []]


def test_postprocess_device_pointer(det, golden_dir):
    import torch
    img = _img(golden_dir, "gt_shrinked_img55.png")
    pred = torch.from_numpy((img.astype(np.float64) / 255.0).astype(np.float32)).reshape(1, 1, 800, 800).cuda()
    torch.cuda.synchronize()
    polys, scores = det.postprocess(pred, 1, 800, 800, np.array([[1.0, 1.0]]), mem_kind=capi.MEM_DEVICE)
    assert polys[0] == K.IMG55_POLYS_ADJ1 and scores[0] == K.IMG55_SCORES


# ---------------------------------------------------------------- recognition
def test_rec_logits_labels_probs(rec, rec_w, golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    crops = W.synth_crops(int(g["rec_input_seed"]), 32)
    logits = rec.forward_host(crops)
    assert np.abs(logits - g["rec_logits"]).max() < TOL
    labels, probs = rec.classify_host(crops)
    assert labels.tolist() == g["rec_labels"].tolist()          # bit-exact label indices
    assert np.abs(probs - g["rec_probs"]).max() < 1e-5


def test_rec_batch256_matches_oracle(rec, rec_w):
    crops = W.synth_crops(2, 256)                                 # BASELINE config 2 shape
    ref_logits = T.rec_forward(rec_w, crops)
    ref_labels, ref_probs = T.rec_classify(ref_logits)
    logits = rec.forward_host(crops)
    assert np.abs(logits - ref_logits).max() < TOL
    labels, probs = rec.classify_host(crops)
    srt = np.sort(ref_logits, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert decided.mean() > 0.95
    assert (labels[decided] == ref_labels[decided]).all()
    top2 = np.argsort(ref_logits, axis=1)[:, -2:]
    assert all(labels[i] in top2[i] for i in np.flatnonzero(~decided))   # a near-tie may go either way, never to a third class
    assert np.abs(probs - ref_probs).max() < 1e-5
    assert len(set(labels.tolist())) > 5


def test_rec_single_crop_and_empty(rec, rec_w):
    crops = W.synth_crops(4, 1)
    assert rec.classify_host(crops)[0].tolist() == T.rec_classify(T.rec_forward(rec_w, crops))[0].tolist()
    labels, probs = rec.classify_host(np.zeros((0, 784), np.float32))
    assert labels.shape == (0,)


@pytest.mark.parametrize("n", [2, 3, 5, 768, 769, 3075])
def test_rec_ragged_batches_cover_every_kernel_variant(rec, rec_w, n):
    """The conv stage takes 1, 2 or 4 crops per workgroup depending on the batch (rec_net.hip) and the last
    workgroup / GEMM tile is partial for these sizes: every crop against the oracle, none written past the end."""
    crops = W.synth_crops(11 + n, n)
    ref = T.rec_forward(rec_w, crops)
    logits = rec.forward_host(crops)
    assert logits.shape == (n, 62) and np.abs(logits - ref).max() < TOL
    labels, probs = rec.classify_host(crops)
    rl, rp = T.rec_classify(ref)
    srt = np.sort(ref, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert (labels[decided] == rl[decided]).all() and np.abs(probs - rp).max() < 1e-5
    top2 = np.argsort(ref, axis=1)[:, -2:]
    assert all(labels[i] in top2[i] for i in np.flatnonzero(~decided))


def test_rec_device_path_guards_and_profile(rec, rec_w):
    import torch
    n = 1001
    crops = W.synth_crops(5, n)
    d = torch.from_numpy(crops).cuda()
    labels = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")     # canaries behind the last crop
    probs = torch.full((n + 8,), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    prof = rec.classify_profile(d.data_ptr(), n, labels.data_ptr(), probs.data_ptr())
    rec.synchronize()
    names = [p[0] for p in prof]
    assert names == ["rec_conv_small_x3", "rec_fc1_ksplit", "rec_fc2_small_softmax_top1"]     # n <= 1024: the latency kernels
    big = rec.classify_profile(torch.from_numpy(W.synth_crops(6, 2000)).cuda().data_ptr(), 2000)
    assert [p[0] for p in big] == ["rec_conv<2>", "rec_fc1", "rec_fc2_softmax_top1"]
    assert all(ms > 0 for _, ms, _, _ in prof)
    assert abs(sum(fl for _, _, fl, _ in prof) / n / 8.587264e6 - 1.0) < 0.05   # executes the reference graph's work (+ K / N padding)
    rl, _ = T.rec_classify(T.rec_forward(rec_w, crops))
    assert labels[:n].cpu().tolist() == rl.tolist()
    assert (labels[n:] == -7).all() and (probs[n:] == -7.0).all()


def test_composed_fpn_matches_layerwise_graph(det, det_w, monkeypatch):
    """The default engine folds in2/in3 into out2/out3 (lateral conv + phase convs on the low-res grid) and
    splits bin_conv1 over the concat into phase convs.  Exact in real arithmetic;
    in f32 it re-associates sums, so it is held to the same bars as everything else: p2 / p3 and the
    bin_conv1 output against the oracle's activations, the map within TOL of the oracle and of the
    layer-by-layer engine (option fpn_unfused=1)."""
    n, h, w = 2, 96, 160
    x = W.synth_image_batch(21, n, h, w)
    st = {}
    ref = T.det_forward(det_w, x, st)
    prob = det.forward_host(x)
    fuse = st["fuse"]                                   # N x 256 x H/4 x W/4 = cat[p5, p4, p3, p2]
    p2 = det.debug_stage(9, (n, h // 4, w // 4, 64))     # returned as NCHW
    p3 = det.debug_stage(10, (n, h // 8, w // 8, 64))
    b1 = det.debug_stage(13, (n, h // 4, w // 4, 64))
    assert _rel(p2, fuse[:, 192:256]) < 2e-5
    assert _rel(p3, fuse[:, 128:192, ::2, ::2]) < 2e-5
    assert _rel(b1, st["bin1"]) < 2e-5
    assert np.abs(prob - ref).max() < TOL
    plain = capi.Detector(W.pack_blob(det_w), 0, options="fpn_unfused=1")
    try:
        prob_plain = plain.forward_host(x)
        sum2 = plain.debug_stage(5, (n, h // 4, w // 4, 256))     # only the layer-wise graph materialises it
        assert sum2.shape[1] == 256
    finally:
        plain.close()
    d = float(np.abs(prob - prob_plain).max())
    print(f"composed vs layer-wise FPN: max |dp| = {d:.3e}")
    assert d < 1e-5
    with pytest.raises(capi.OcrError):
        det.debug_stage(5, (n, h // 4, w // 4, 256))


def test_stem_exact_and_inexact_tiles(det, det_w):
    """The split-bf16 stem skips the products of the mid / lo operand planes in tiles whose pixels are all exactly bf16 (raw luma) and
    runs all six elsewhere, decided per tile: a frame that is integer-valued on the left and has fractional pixels on the right (plus
    one lone fractional pixel in the integer part) against the oracle and against the exact-f32 engine."""
    n, h, w = 2, 128, 192
    x = W.synth_image_batch(61, n, h, w)
    rng = np.random.default_rng(61)
    x[:, :, :, w // 2:] += rng.random((n, 1, h, w - w // 2), dtype=np.float32)            # 24-bit mantissas
    x[0, 0, 37, 11] += np.float32(0.3)
    got = det.forward_host(x)
    assert np.abs(got - T.det_forward(det_w, x)).max() < TOL
    f32 = capi.Detector(W.pack_blob(det_w), 0, options="mfma=f32")
    try:
        assert np.abs(got - f32.forward_host(x)).max() < 1e-5
    finally:
        f32.close()
    stem = det.debug_stage(0, (n, h // 4, w // 4, 64))
    stages = {}
    T.det_forward(det_w, x, stages)
    ref = stages["stem"]
    if stem.shape != ref.shape:
        stem = np.transpose(stem, (0, 3, 1, 2))
    assert np.abs(stem - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("options", [
    "winograd_fused=0",                                            # direct convs on the large grids, unfused Winograd layer3/4
    "winograd=0;winograd_fused=0",                                 # no Winograd at all
    "bin_pyr=0",                                                   # bin_conv1 as four launches
    "phase_windows=0",                                             # the FPN's up-2 phase convs as one 64-column tile per phase (default: rows = 2 x 2 windows)
    "pyr_grouped=0",                                               # bin_conv1 over p5..p3 as one 64-column tile per phase (default: phase blocks)
    "fpn_unfused=1",                                               # layer-by-layer FPN
    "fpn_unfused=1;winograd=0;winograd_fused=0;tail_unfused=1",    # the plain graph
    "overlap=1", "overlap=2",                                      # second-stream schedules
    "transform_fuse=1",                                            # layer3 / layer4, block 1: output transform of conv1 + input transform of conv2 in one launch
    "overlap=0", "overlap=3;w43_side_cus=128",                     # one stream (the default is 3: FPN Winograd launches + bin_conv1 p2 term beside layer3 / layer4)
    "mfma=f32",                                                    # every conv on the exact-f32 MFMA (no split-bf16 kernels)
    "mfma=f32;bin_pyr=0",
    "winograd43_x3=1",                                             # the fused F(4x4) convs on the bf16 matrix cores too (winograd43_x3.hip)
    "out4_fused=1",                                                # out4 on the fused kernel, out5 as a direct conv
    "winograd43_x3=1;out4_fused=1",                                # ... out4 through the 256-channel instantiation of winograd43_x3.hip
    "x3_wide=1",                                                   # the split-bf16 NHWC convs on the 256 x 128 persistent form (conv_x3w.hip)
    "x3_wide=1;overlap=0",
])
def test_engine_modes_agree(det, det_w, options):
    """Every graph-level option of the engine (ocr_det_create_with_options) computes the same
    map: within 1e-5 of the default engine and within TOL of the oracle."""
    n, h, w = 2, 96, 160
    x = W.synth_image_batch(33, n, h, w)
    base = det.forward_host(x)
    ref = T.det_forward(det_w, x)
    other = capi.Detector(W.pack_blob(det_w), 0, options=options)
    try:
        got = other.forward_host(x)
    finally:
        other.close()
    assert np.abs(got - base).max() < 1e-5
    assert np.abs(got - ref).max() < TOL


def test_fused_transforms_are_bit_identical(det_w):
    """winograd43_out_in_kernel (layer3 / layer4: M -> y -> V of two neighbouring 3x3 convs in one launch, the activation in LDS) does the
    arithmetic of the two separate transform kernels operation for operation: the same map, bit for bit - on a frame size whose deep
    grids are ragged (H/16 = 6, W/16 = 10: partial 4 x 4 tiles) and on one whose are not, one stream so that nothing else re-associates."""
    for (n, h, w) in ((3, 96, 160), (2, 128, 192), (1, 64, 64)):
        x = W.synth_image_batch(41, n, h, w)
        a = capi.Detector(W.pack_blob(det_w), 0, options="overlap=0;transform_fuse=1")
        b = capi.Detector(W.pack_blob(det_w), 0, options="overlap=0;transform_fuse=0")
        try:
            assert np.array_equal(a.forward_host(x), b.forward_host(x)), (n, h, w)
        finally:
            a.close()
            b.close()


def test_wide_form_and_oversubscribed_grids_are_bit_identical(det_w):
    """Which workgroup computes which tile is not part of the result: the split-bf16 NHWC convs on the 256 x 128 persistent form against the
    128-wide tiles, and the fused Winograd launches on twice as many workgroups as the chip holds (head_cus_yield=2: what the pipelined calls
    use while the previous batch's polygon chain shares the CUs) - the same map bit for bit, on a configs[1]-shaped batch and a ragged small one."""
    for (n, h, w) in ((8, 640, 640), (3, 96, 160)):
        x = W.synth_image_batch(47, n, h, w)
        outs = []
        for opt in ("overlap=0;x3_wide=0", "overlap=0;x3_wide=1", "overlap=0;w43_cus=512", "overlap=0;w43_cus=96"):
            d = capi.Detector(W.pack_blob(det_w), 0, options=opt)
            try:
                outs.append((opt, d.forward_host(x)))
            finally:
                d.close()
        for opt, o in outs[1:]:
            assert np.array_equal(o, outs[0][1]), (opt, n, h, w)


def test_phase_blocks_are_bit_identical(det_w):
    """bin_conv1 over p5, p4, p3 (conv_igemm.hip, PYRG): phases that read the same source rows are column groups of one 128-wide tile
    (+ a launch for the four corner phases).  Per output element the same products in the same order as the one-tile-per-phase form:
    the same map bit for bit - with one cell block (plain tile order), with a partial last cell block in chunked order (400 cells),
    and on a ragged p5 grid; one stream and the default schedule; the split-bf16
    kernel and the bf16 kernel."""
    for (n, h, w) in ((2, 96, 160), (4, 320, 320), (3, 224, 352)):
        x = W.synth_image_batch(43, n, h, w)
        for sched in ("overlap=0", "overlap=3", "precision=bf16;overlap=0", "precision=bf16;overlap=3"):
            a = capi.Detector(W.pack_blob(det_w), 0, options=sched + ";pyr_grouped=1")
            b = capi.Detector(W.pack_blob(det_w), 0, options=sched + ";pyr_grouped=0")
            try:
                assert np.array_equal(a.forward_host(x), b.forward_host(x)), (n, h, w, sched)
            finally:
                a.close()
                b.close()


def test_window_indexed_phase_convs_are_bit_identical(det_w):
    """The FPN's up-2 phase convs (conv_igemm.hip, WING): rows of the GEMM are the (H + 1) x (W + 1) 2 x 2 windows of the low-res grid, the four
    phases that read a window are four column groups of one operand tile, outputs outside the map are dropped.  Per output the products and
    their order are those of the one-tile-per-phase form: the same map bit for bit - odd and ragged low-res grids (windows past every
    border), one cell row, several row tiles; the composed FPN with one phase launch for bin_conv1 and with four (bin_pyr=0: p3's up-2
    term with a residual); the split-bf16 kernel and the bf16 kernel."""
    for (n, h, w) in ((2, 96, 160), (3, 224, 352), (1, 32, 64), (4, 320, 320)):
        x = W.synth_image_batch(53, n, h, w)
        for sched in ("overlap=0", "overlap=3", "overlap=0;bin_pyr=0", "precision=bf16;overlap=0", "precision=bf16;overlap=3"):
            a = capi.Detector(W.pack_blob(det_w), 0, options=sched + ";phase_windows=1")
            b = capi.Detector(W.pack_blob(det_w), 0, options=sched + ";phase_windows=0")
            try:
                assert np.array_equal(a.forward_host(x), b.forward_host(x)), (n, h, w, sched)
            finally:
                a.close()
                b.close()


def test_engine_options_are_explicit_and_checked(det_w, monkeypatch):
    """The library never reads the environment (a maintainer's shell cannot silently select another engine); unknown
    or malformed options are errors."""
    x = W.synth_image_batch(8, 1, 64, 64)
    base = capi.Detector(W.pack_blob(det_w), 0)
    want = base.forward_host(x)
    base.close()
    for k, v in (("OCR_WINOGRAD_FUSED", "0"), ("OCR_FPN_UNFUSED", "1"), ("OCR_DET_PRECISION", "bf16"), ("OCR_TAIL_UNFUSED", "1")):
        monkeypatch.setenv(k, v)
    d = capi.Detector(W.pack_blob(det_w), 0)
    assert np.array_equal(d.forward_host(x), want)      # bit for bit the default engine
    d.close()
    b16 = capi.Detector(W.pack_blob(det_w), 0, options="precision=bf16")
    assert not np.array_equal(b16.forward_host(x), want)
    b16.close()
    for bad in ("winograd_fused", "no_such_option=1", "precision=fp8", "overlap=x"):
        with pytest.raises(capi.OcrError) as e:
            capi.Detector(W.pack_blob(det_w), 0, options=bad)
        assert e.value.code == 1


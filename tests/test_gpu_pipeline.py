"""BASELINE config 0 plumbing on the GPU: preprocessed frame -> detect -> polygons -> crops -> recognise,
checked stage by stage against the oracles (the crop rule is build-defined: oracle/crop_oracle.py)."""
import os

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import crop_oracle as CR
from oracle import postproc_oracle as O
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu


def test_crops_match_oracle_on_text_like_maps(golden_dir):
    det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
    rng = np.random.RandomState(0)
    frames = rng.randint(0, 256, (2, 1, 800, 800)).astype(np.float32)
    prob = np.stack([np.array(Image.open(os.path.join(golden_dir, f)).convert("L")).astype(np.float32) / 255.0
                     for f in ("gt_shrinked_img55.png", "gt_shrinked_img545.png")])[:, None]
    adj = np.array([[800 / 300, 533 / 200], [1.0, 1.0]])
    polys, scores, crops = det.postprocess_and_crops(prob, frames, adj)
    opolys, oscores = O.get_boxes_and_box_scores(prob, adj)
    assert polys == opolys and scores == oscores and len(polys[0]) == 4
    ocrops = CR.extract_crops(frames, opolys, adj)
    assert crops.shape == ocrops.shape == (sum(len(p) for p in polys), 784)
    assert np.array_equal(crops, ocrops)
    assert 0.0 <= crops.min() and crops.max() <= 1.0
    det.close()


def test_detect_crop_recognise_end_to_end(golden_dir):
    """preprocessed_img55.png -> forward -> polygons (text-like map of the fixture) -> crops -> labels."""
    det_w, rec_w = W.make_det_weights(0), W.make_rec_weights(0)
    det = capi.Detector(W.pack_blob(det_w), 0)
    rec = capi.Recognizer(W.pack_blob(rec_w), 0)
    frame = np.array(Image.open(os.path.join(golden_dir, "preprocessed_img55.png")).convert("L")).astype(np.float32)
    frames = frame.reshape(1, 1, 800, 800)
    pred = det.forward_host(frames)
    assert np.abs(pred - T.det_forward(det_w, frames)).max() < 1e-4
    # random weights give a noise map; use the fixture's own text mask as the probability map for the
    # geometry leg and the real frame for the crops
    prob = (np.array(Image.open(os.path.join(golden_dir, "gt_shrinked_img55.png")).convert("L")).astype(np.float32)
            / 255.0).reshape(1, 1, 800, 800)
    adj = np.array([[800 / 300, 533 / 200]])                      # image_ops.rs:910-916
    polys, scores, crops = det.postprocess_and_crops(prob, frames, adj)
    assert len(polys[0]) == 4
    ocrops = CR.extract_crops(frames, polys, adj)
    assert np.array_equal(crops, ocrops)
    labels, probs = rec.classify_host(crops)
    ref_logits = T.rec_forward(rec_w, ocrops)
    rl, rp = T.rec_classify(ref_logits)
    srt = np.sort(ref_logits, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert (labels[decided] == rl[decided]).all() and np.abs(probs - rp).max() < 1e-5
    det.close()
    rec.close()

"""The plain-C CNN oracle vs ATen (torch CPU) vs the committed goldens."""
import os

import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import weights as W
from oracle import cnn_oracle as C
from oracle import torch_ref as T

TOL = 1e-4  # BASELINE.json north_star: fp32 probability maps within 1e-4


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz"))


def test_param_inventory():
    p = W.make_det_weights(0)
    assert len(p) == 121 and sum(v.size for v in p.values()) == 12_180_097
    r = W.make_rec_weights(0)
    assert sum(v.size for v in r.values()) == 608_702
    blob = W.pack_blob(p)
    q = W.unpack_blob(blob)
    assert list(q) == list(p) and all(np.array_equal(q[k], p[k]) for k in p)


def test_det_c_oracle_matches_golden(gold):
    p = W.make_det_weights(int(gold["det_seed"]))
    x = W.synth_image_batch(int(gold["det_input_seed"]), 2, 64, 96)
    st = {}
    prob = C.det_forward(p, x, st)
    assert prob.shape == (2, 1, 64, 96)
    assert np.abs(prob - gold["det_prob"]).max() < TOL
    assert np.abs(st["logit"] - gold["det_logit"]).max() < TOL
    assert np.abs(st["stem"][:, :8] - gold["det_stem"]).max() < TOL
    assert np.abs(st["layer4"][:, :16] - gold["det_layer4"]).max() < 5e-4   # activations up to ~30


def test_det_torch_matches_golden(gold):
    p = W.make_det_weights(0)
    x = W.synth_image_batch(7, 2, 64, 96)
    assert np.abs(T.det_forward(p, x) - gold["det_prob"]).max() < 1e-5


def test_det_oracles_agree_on_ragged_shape():
    p = W.make_det_weights(3)
    x = W.synth_image_batch(11, 1, 32, 96)    # smallest legal height (multiple of 32)
    assert np.abs(C.det_forward(p, x) - T.det_forward(p, x)).max() < TOL


def test_rec_oracles_match_golden(gold):
    r = W.make_rec_weights(0)
    crops = W.synth_crops(2, 32)
    for impl in (C, T):
        logits = impl.rec_forward(r, crops)
        assert np.abs(logits - gold["rec_logits"]).max() < TOL
        lab, pr = impl.rec_classify(logits)
        assert lab.tolist() == gold["rec_labels"].tolist()
        assert np.abs(pr - gold["rec_probs"]).max() < 1e-5
    assert len(set(gold["rec_labels"].tolist())) > 3


def test_alphabet():
    assert len(T.VALUES) == 62 and T.VALUES[0] == "A" and T.VALUES[26] == "a" and T.VALUES[61] == "9"

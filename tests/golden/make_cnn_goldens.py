"""Generates tests/golden/cnn_goldens.npz from oracle/torch_ref.py (ATen CPU, the
operator library the reference reaches through tch).  The reference itself holds
no CNN fixture, so these vectors pin our own oracle and kernels to
each other, not to the reference.  Run from the repo root:
    python tests/golden/make_cnn_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import weights as W  # noqa: E402
from oracle import torch_ref as T  # noqa: E402

import torch  # noqa: E402

torch.set_num_threads(1)  # fixed summation order
det = W.make_det_weights(0)
rec = W.make_rec_weights(0)
x = W.synth_image_batch(7, 2, 64, 96)
st = {}
prob = T.det_forward(det, x, st)
crops = W.synth_crops(2, 32)
logits = T.rec_forward(rec, crops)
labels, probs = T.rec_classify(logits)
np.savez_compressed(
    os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"),
    det_seed=0, det_input_seed=7, det_prob=prob.astype(np.float32),
    det_stem=st["stem"][:, :8].astype(np.float32), det_layer4=st["layer4"][:, :16].astype(np.float32),
    det_logit=st["logit"].astype(np.float32),
    rec_seed=0, rec_input_seed=2, rec_logits=logits.astype(np.float32), rec_labels=labels, rec_probs=probs)
print("prob", prob.shape, float(prob.mean()), "labels", labels.tolist())

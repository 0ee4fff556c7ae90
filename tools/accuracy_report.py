#!/usr/bin/env python3
"""Distance of the HIP detector from the ATen-CPU oracle under each graph-level engine option (run on the GPU box):
python tools/accuracy_report.py [n] [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi, weights as W  # noqa: E402
from oracle import torch_ref as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = W.make_det_weights(0)
x = W.synth_image_batch(5, n, s, s)
ref = T.det_forward(w, x)
for label, options in (("default (composed FPN, fused Winograd, single-launch bin_conv1)", None),
                       ("mfma=f32 (every conv on the exact-f32 matrix instructions, no split-bf16 kernels)", "mfma=f32"),
                       ("winograd43=0 (layer3 / layer4 as F(2x2,3x3))", "winograd43=0"),
                       ("winograd_fused=0", "winograd_fused=0"),
                       ("direct convs, composed FPN", "winograd=0;winograd_fused=0"),
                       ("fpn_unfused=1", "fpn_unfused=1"),
                       ("layer-by-layer direct convs (the graph as model.rs writes it)", "winograd=0;winograd_fused=0;fpn_unfused=1;tail_unfused=1"),
                       ("the same, exact-f32 matrix instructions only", "mfma=f32;winograd=0;winograd_fused=0;fpn_unfused=1;tail_unfused=1")):
    det = capi.Detector(W.pack_blob(w), 0, options=options)
    d = np.abs(det.forward_host(x) - ref)
    det.close()
    print(f"{label:70s} max |dprob| {d.max():.3e}  mean {d.mean():.3e}")

#!/usr/bin/env python3
"""Distance of the HIP detector from the ATen-CPU oracle under each graph-level option (run on the GPU box):
python tools/accuracy_report.py [n] [size]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import ocr_rs_amd  # noqa: F401
    from ocr_rs_amd import capi, weights as W
    from oracle import torch_ref as T
    n, s = int(sys.argv[2]), int(sys.argv[3])
    w = W.make_det_weights(0)
    x = W.synth_image_batch(5, n, s, s)
    ref = T.det_forward(w, x)
    det = capi.Detector(W.pack_blob(w), 0)
    got = det.forward_host(x)
    d = np.abs(got - ref)
    print(f"max |dprob| {d.max():.3e}  mean {d.mean():.3e}")
    sys.exit(0)

n = sys.argv[1] if len(sys.argv) > 1 else "2"
s = sys.argv[2] if len(sys.argv) > 2 else "256"
for label, env in (("default (composed FPN, fused Winograd layer1/2, Winograd layer3/4)", {}),
                   ("OCR_WINOGRAD_FUSED=0", {"OCR_WINOGRAD_FUSED": "0"}),
                   ("OCR_WINOGRAD=0 OCR_WINOGRAD_FUSED=0 (direct convs, composed FPN)", {"OCR_WINOGRAD": "0", "OCR_WINOGRAD_FUSED": "0"}),
                   ("OCR_FPN_UNFUSED=1", {"OCR_FPN_UNFUSED": "1"}),
                   ("OCR_WINOGRAD=0 OCR_WINOGRAD_FUSED=0 OCR_FPN_UNFUSED=1 (layer-by-layer direct convs)",
                    {"OCR_WINOGRAD": "0", "OCR_WINOGRAD_FUSED": "0", "OCR_FPN_UNFUSED": "1"})):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, s], env=e, capture_output=True, text=True)
    print(f"{label:88s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr.strip()[-200:]}")

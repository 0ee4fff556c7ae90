#!/usr/bin/env python3
"""tch VarStore file -> OCRW blob for ocr_det_create / ocr_rec_create:
python tools/import_varstore.py <model.ot> <out.ocrw> det|rec"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import weights as W  # noqa: E402

if len(sys.argv) != 4 or sys.argv[3] not in ("det", "rec"):
    sys.exit(__doc__)
params = W.load_varstore(sys.argv[1], kind=sys.argv[3])
blob = W.pack_blob(params)
open(sys.argv[2], "wb").write(blob)
print(f"{sys.argv[2]}: {len(params)} tensors, {len(blob)} bytes")

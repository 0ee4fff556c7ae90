"""Split-bf16 conv launch shapes of the detector under one (compile-time ablated) library:
   OCR_AMD_LIB=ocr-rs_amd/lib_abl<bits>/libocr_amd.so python3 tools/x3_abl_shapes.py <label>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
SHAPES = [  # name, n, h, w, cin, cout, ks, stride
    ("layer2.0.conv1", 32, 160, 160, 64, 128, 3, 2),
    ("layer3.0.conv1", 32, 80, 80, 128, 256, 3, 2),
    ("layer4.0.conv1", 32, 40, 40, 256, 512, 3, 2),
    ("wino gemm l3", 36, 40, 80, 256, 256, 1, 1),
    ("wino gemm l4", 36, 20, 40, 512, 512, 1, 1),
]
label = sys.argv[1] if len(sys.argv) > 1 else ""
for name, n, h, w, cin, cout, ks, st in SHAPES:
    ms = min(det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64, 10) for _ in range(3))
    print(f"{label:12s} {name:18s} {ms:.4f} ms", flush=True)

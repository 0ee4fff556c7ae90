"""conv_igemm's 128-wide split-bf16 tiles against the 256 x 128 persistent form (conv_x3w.hip) on the detector's launch shapes, random operands.
   python3 tools/bench_x3w.py"""
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
    ("layer2.0.down", 32, 160, 160, 64, 128, 1, 2),
    ("layer3.0.down", 32, 80, 80, 128, 256, 1, 2),
    ("layer4.0.down", 32, 40, 40, 256, 512, 1, 2),
    ("wino gemm l3 (one B)", 36, 40, 80, 256, 256, 1, 1),
    ("wino gemm l4 (one B)", 36, 20, 40, 512, 512, 1, 1),
    ("in5", 32, 20, 20, 512, 256, 1, 1),
    ("in4", 32, 40, 40, 256, 256, 1, 1),
]
for rep in range(2):
    for name, n, h, w, cin, cout, ks, st in SHAPES:
        pad = (ks - 1) // 2
        ho, wo = (h + 2 * pad - ks) // st + 1, (w + 2 * pad - ks) // st + 1
        gf = 2.0 * n * ho * wo * cout * ks * ks * cin / 1e9
        a = det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64, 20)
        b = det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64 | 128, 20)
        print(f"{name:22s} {gf:7.2f} GF  128-wide {a:.4f} ms ({6 * gf / a / 2500:.3f} of 2.5 PF) | 256x128 {b:.4f} ms ({6 * gf / b / 2500:.3f})  ratio {b / a:.3f}", flush=True)

"""f32 conv_igemm against its split-bf16 form on the detector's MFMA-bound launch shapes (random operands).
   python3 tools/bench_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
SHAPES = [  # name, n, h, w, cin, cout, ks, stride
    ("layer2.0.conv1", 32, 160, 160, 64, 128, 3, 2),
    ("layer3.0.conv1", 32, 80, 80, 128, 256, 3, 2),
    ("layer4.0.conv1", 32, 40, 40, 256, 512, 3, 2),
    ("wino gemm l3", 36, 40, 80, 256, 256, 1, 1),
    ("wino gemm l4", 36, 20, 40, 512, 512, 1, 1),
    ("3x3 s1 128", 32, 80, 80, 128, 128, 3, 1),
    ("3x3 s1 256->64", 32, 40, 40, 256, 64, 3, 1),
]
for tile in (0, 1, 2):
    capi.test_lib().ocr_test_set_conv_tile(tile)
    for name, n, h, w, cin, cout, ks, st in SHAPES:
        pad = (ks - 1) // 2
        ho, wo = (h + 2 * pad - ks) // st + 1, (w + 2 * pad - ks) // st + 1
        gf = 2.0 * n * ho * wo * cout * ks * ks * cin / 1e9
        a = det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32, 10) if tile == 0 else float("nan")
        b = det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64, 10)
        print(f"tile {tile} {name:18s} {gf:7.2f} GF  f32 {a:.4f} ms {gf / a:7.1f} TF/s | x3 {b:.4f} ms {gf / b:7.1f} TF/s (bf16 MFMA {6 * gf / b:7.1f})", flush=True)

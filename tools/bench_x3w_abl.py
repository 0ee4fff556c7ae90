"""One line per launch shape: the wide split-bf16 form's time in this library (tools/build_abl_x3w.sh variants via OCR_AMD_LIB).
   OCR_AMD_LIB=ocr-rs_amd/lib_x3w<N>/libocr_amd.so python3 tools/bench_x3w_abl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
SHAPES = [("layer2.0.conv1", 32, 160, 160, 64, 128, 3, 2), ("layer3.0.conv1", 32, 80, 80, 128, 256, 3, 2), ("wino gemm l3", 36, 40, 80, 256, 256, 1, 1),
          ("wino gemm l4", 36, 20, 40, 512, 512, 1, 1)]
out = []
for name, n, h, w, cin, cout, ks, st in SHAPES:
    t = min(det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64 | 128, 20) for _ in range(2))
    out.append(f"{name} {t:.4f}")
print(os.environ.get("OCR_AMD_LIB", "default"), " | ".join(out), flush=True)

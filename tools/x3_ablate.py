"""Ablations of the split-bf16 conv kernel (diagnostic build):
   make -C ocr-rs_amd/csrc EXTRA=-DIGEMM_DEBUG OUT=../lib_dbg && OCR_AMD_LIB=ocr-rs_amd/lib_dbg/libocr_amd.so python3 tools/x3_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library

det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
T = capi.test_lib()
SHAPES = [("layer3.0.conv1", 32, 80, 80, 128, 256, 3, 2), ("wino gemm l3", 36, 40, 80, 256, 256, 1, 1)]
NAMES = {0: "full", 128: "full, no epilogue", 111: "loop overhead", 111 | 128: "loop overhead, no epilogue", 111 | 256: "one iteration", 111 | 128 | 256: "one iteration, no epilogue"}
for tile in (1, 2):
    T.ocr_test_set_conv_tile(tile)
    for name, n, h, w, cin, cout, ks, st in SHAPES:
        for dbg, label in NAMES.items():
            T.ocr_test_set_conv_debug(dbg)
            ms = det.debug_conv_bench(n, h, w, cin, cout, ks, st, 32 | 64, 10)
            print(f"tile {tile} {name:16s} {label:34s} {ms:.4f} ms", flush=True)
T.ocr_test_set_conv_debug(0)

"""One split-bf16 conv launch shape, a few launches (for rocprofv3 counter passes): python3 tools/x3_one.py [tile] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()   # the hooks below set library-wide state: detector and hooks from one library
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
capi.test_lib().ocr_test_set_conv_tile(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
mode = 32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else 32 | 64
print(det.debug_conv_bench(32, 80, 80, 128, 256, 3, 2, mode, 5))
print(det.debug_conv_bench(36, 40, 80, 256, 256, 1, 1, mode, 5))

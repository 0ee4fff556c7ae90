"""Two complementary kernels on two streams (run on the GPU box): the fused F(4x4,3x3) kernel (f32 MFMA, bf16 matrix cores idle) beside
one split-bf16 conv_igemm launch (bf16 MFMA).  Prints T_A, T_B per launch, the one-stream time of the two queues, the two-stream
wall time and their ratio.   python3 tools/dual_stream.py [frames]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names = {0: "igemm_x3 3x3 s2 64->128 @160", 1: "igemm_x3 3x3 s2 128->256 @80", 2: "igemm_x3 batched 36 GEMMs K=256 (layer3)",
         3: "igemm_x3 3x3 s2 256->512 @40"}
L = capi.test_lib()
print(f"frames {n}; A = winograd43_fused<c64> on {n}x160x160x64")
print(f"{'B':42s} {'A grid (CUs)':>12s} {'T_A ms':>8s} {'T_B ms':>8s} {'repsA':>5s} {'repsB':>5s} {'serial':>8s} {'2-stream':>8s} {'ratio':>6s}")
for which in (1, 0, 2, 3):
    for cus in (256, 128):
        ms = (C.c_float * 4)()
        capi.check(L.ocr_test_dual_stream_bench(det._h, n, which, cus, 4, 4, ms))
        ta, tb = ms[0], ms[1]
        # balance the queues: about 4 ms of each
        ra, rb = max(2, round(4.0 / ta)), max(2, round(4.0 / tb))
        capi.check(L.ocr_test_dual_stream_bench(det._h, n, which, cus, ra, rb, ms))
        print(f"{names[which]:42s} {cus:12d} {ms[0]:8.4f} {ms[1]:8.4f} {ra:5d} {rb:5d} {ms[3]:8.3f} {ms[2]:8.3f} {ms[2] / ms[3]:6.3f}", flush=True)

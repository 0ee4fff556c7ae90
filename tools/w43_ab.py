"""A/B of the fused F(4x4,3x3) kernel in ONE process (devices differ by several per cent), alternating forwards over
settings of the test hook's word: 0 = shipped (XCD-contiguous runs), 16 = plain linear block order.
python3 tools/w43_ab.py [rounds] [word ...]
(Round 3 also tried starting the second half of the grid 8 k ... 65 k cycles late, so that the two workgroups of a CU run
out of phase: 0.2142 -> 0.2143 / 0.2163 / 0.2186 / 0.2233 ms - the delay is only partly won back, removed.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
NB = int(os.environ.get("W43_BATCH", "32"))
x = torch.from_numpy(W.synth_image_batch(1, NB, 640, 640)).cuda()
prob = torch.empty_like(x)
VARIANTS = [int(v, 0) for v in sys.argv[2:]] or [0, 16]
acc = {v: {} for v in VARIANTS}
for r in range(rounds + 1):
    for dbg in VARIANTS:
        capi.test_lib().ocr_test_w43_debug(dbg)
        for nm, ms, fl, by in det.forward_profile(x.data_ptr(), NB, 640, 640, prob.data_ptr()):
            if r and "winograd43_fused" in nm:
                e = acc[dbg].setdefault(nm, [0.0, 0])
                e[0] += ms
                e[1] += 1
capi.test_lib().ocr_test_w43_debug(0)
for nm in acc[VARIANTS[0]]:
    print(f"{nm:28s} " + "   ".join(f"[{v:#x}] {acc[v][nm][0] / acc[v][nm][1]:.4f} ms" for v in VARIANTS))

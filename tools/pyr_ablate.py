"""Ablations of the split-bf16 phase convs inside a whole forward (diagnostic build, per-launch HIP events):
   make -C ocr-rs_amd/csrc EXTRA=-DIGEMM_DEBUG OUT=../lib_dbg && OCR_AMD_LIB=ocr-rs_amd/lib_dbg/libocr_amd.so python3 tools/pyr_ablate.py [engine options]
bits: 1 no A DMA, 2 no B DMA, 4 no split, 8 no MFMA, 32 no LDS fragment reads, 64 no mid-step barrier, 128 no epilogue, 256 one K-step,
512 no wait for the DMA (the results of an ablated run are garbage; only the times mean something)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocr_rs_amd  # noqa
from ocr_rs_amd import capi, weights as W
capi.use_test_library()

n, s = 32, 640
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options=sys.argv[1] if len(sys.argv) > 1 else None)
T = capi.test_lib()
x = torch.from_numpy(W.synth_image_batch(1, n, s, s)).cuda()
prob = torch.empty_like(x)
torch.cuda.synchronize()
CASES = [(0, "full"), (1, "no A DMA"), (2, "no B DMA"), (3, "no DMA"), (4, "no split"), (8, "no MFMA"), (12, "no split, no MFMA"), (32, "no LDS reads"),
         (64, "no mid barrier"), (512, "no DMA wait"), (128, "no epilogue"), (256, "one K-step"), (256 | 128, "one K-step, no epilogue"),
         (1 | 2 | 4 | 8 | 32 | 64, "loop overhead"), (1 | 2 | 4 | 8 | 32 | 64 | 128, "loop overhead, no epilogue")]
rows = {}
for dbg, label in CASES:
    T.ocr_test_set_conv_debug(dbg)
    acc = None
    for r in range(4):
        prof = det.forward_profile(x.data_ptr(), n, s, s, prob.data_ptr())
        if r == 0:
            continue
        if acc is None:
            acc = [[nm, 0.0] for nm, ms, fl, by in prof]
        for i, (nm, ms, fl, by) in enumerate(prof):
            acc[i][1] += ms / 3
    for i, (nm, ms) in enumerate(acc):
        if "PYR4" in nm or "PHASE2" in nm or (",k3,s2," in nm and i < 12):
            rows.setdefault((i, nm), []).append(ms)
T.ocr_test_set_conv_debug(0)
print(f"{'launch':52s}" + "".join(f"{lab[:13]:>14s}" for _, lab in CASES))
for (i, nm), v in rows.items():
    print(f"{i:2d} {nm:49s}" + "".join(f"{ms:14.4f}" for ms in v))

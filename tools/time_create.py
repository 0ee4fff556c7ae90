import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
import torch; torch.cuda.init()
blob = W.pack_blob(W.make_det_weights(0)); rblob = W.pack_blob(W.make_rec_weights(0))
for opt in (None, "precision=bf16", "mfma=f32"):
    t0 = time.perf_counter(); d = capi.Detector(blob, 0, options=opt); t1 = time.perf_counter()
    print(f"ocr_det_create({opt}): {t1 - t0:.3f} s"); d.close()
t0 = time.perf_counter(); r = capi.Recognizer(rblob, 0); print(f"ocr_rec_create: {time.perf_counter() - t0:.3f} s"); r.close()

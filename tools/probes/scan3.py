import os, sys
sys.path.insert(0, os.getcwd())
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
tag = os.environ.get("OCR_CONV_EXTRA_LDS", "0")
print("extra_lds", tag, " ".join(f"{det.debug_conv_bench(32,160,160,ci,64,3,1,0,10)*1e3:7.1f}" for ci in (64,256)),
      "| 40x40 Cout256 (64x64):", " ".join(f"{det.debug_conv_bench(32,40,40,ci,256,3,1,0,10)*1e3:7.1f}" for ci in (256,)))

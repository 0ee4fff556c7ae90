import os, sys
sys.path.insert(0, os.getcwd())
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
for n in (16, 32):
    print(os.environ.get("OCR_AMD_LIB","").split("/")[-1], n, " ".join(f"{det.debug_conv_bench(n,160,160,ci,64,3,1,0,10)*1e3:7.1f}" for ci in (64,128,256)))

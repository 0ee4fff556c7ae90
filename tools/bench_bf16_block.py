"""layer1's BasicBlock in the bf16 precision at configs[1]'s size (32 x 160 x 160 x 64): one launch (basic_block_bf16_c64.hip) against
the two conv3x3_bf16_c64 launches, HIP-event time per block over 20 repetitions; the results must be the same bits."""
import sys

import numpy as np

sys.path.insert(0, ".")
import ocr_rs_amd  # noqa: F401,E402
from ocr_rs_amd import capi  # noqa: E402
from ocr_rs_amd import weights as W  # noqa: E402

ABL = len(sys.argv) > 1 and sys.argv[1] == "abl"   # one line for an ablation library (tools/build_abl_bb.sh, OCR_AMD_LIB)
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
rng = np.random.default_rng(1)
for n, h, w in ((32, 160, 160),) if ABL else ((32, 160, 160), (32, 200, 200), (8, 160, 160)):
    x = np.maximum(rng.standard_normal((n, h, w, 64), dtype=np.float32), 0)
    w1 = (rng.standard_normal((64, 9, 64), dtype=np.float32) / 24).astype(np.float32)
    w2 = (rng.standard_normal((64, 9, 64), dtype=np.float32) / 24).astype(np.float32)
    s = (0.5 + rng.random(64, dtype=np.float32))
    b = rng.standard_normal(64, dtype=np.float32)
    if ABL:
        ms1 = min(det.debug_bf16_basic_block(x, w1, w2, s, b, s, b, fused=True, iters=20)[1] for _ in range(3))
        import os
        print(f"{os.environ.get('OCR_AMD_LIB', 'default')}: one launch {ms1:.4f} ms", flush=True)
        continue
    two, ms2 = det.debug_bf16_basic_block(x, w1, w2, s, b, s, b, fused=False, iters=20)
    one, ms1 = det.debug_bf16_basic_block(x, w1, w2, s, b, s, b, fused=True, iters=20)
    gf = 2 * 2.0 * n * h * w * 64 * 576 / 1e9
    print(f"{n} x {h} x {w}: two launches {ms2:.4f} ms ({gf / ms2:.0f} TF/s)   one launch {ms1:.4f} ms ({gf / ms1:.0f} TF/s)   "
          f"same bits: {np.array_equal(one, two)}", flush=True)
det.close()

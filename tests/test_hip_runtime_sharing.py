"""One HIP runtime per process, whichever of torch and libocr_amd.so is loaded first (ocr-rs_amd/capi.py::_share_torch_hip_runtime).
The PyTorch-ROCm wheel links its own libamdhip64.so by file name; the library needs `libamdhip64.so.7`: loaded first and on its own it
used to bring in /opt/rocm's copy, and torch's lazy initialisation in the same process then failed with "No HIP GPUs are available"."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_MAPS = """
import sys
sys.path.insert(0, %r)
import ocr_rs_amd
from ocr_rs_amd import capi
%s
maps = open('/proc/self/maps').read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l)))
"""


@pytest.mark.parametrize("order", ["library first", "torch first"])
def test_one_hip_runtime_is_mapped_whichever_loads_first(order):
    body = "capi.lib(); import torch" if order == "library first" else "import torch; capi.lib()"
    r = subprocess.run([sys.executable, "-c", _MAPS % (ROOT, body)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    libs = eval(r.stdout.strip().splitlines()[-1])
    assert len(libs) == 1, libs


@pytest.mark.gpu
def test_torch_initialises_after_the_library_has_used_the_gpu():
    """... the sequence that used to fail: the library drives the GPU (and walks an error path) before torch has touched it."""
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
import ocr_rs_amd
from ocr_rs_amd import capi, weights as W
det = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
x = W.synth_image_batch(1, 1, 64, 64)
p = det.forward_host(x)
try:
    capi.Detector(W.pack_blob(W.make_det_weights(0)), 0, options="overlap=7")
    raise SystemExit("overlap=7 was accepted")
except capi.OcrError:
    pass
try:
    capi.device_contours(np.zeros((4096, 4096), np.uint8))
except capi.OcrError:
    pass
import torch
torch.cuda.init()
t = torch.from_numpy(np.asarray(p)).cuda()
assert torch.equal((t * 2).cpu(), torch.from_numpy(np.asarray(p) * 2))
print("ok", torch.cuda.device_count())
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok"), r.stdout + r.stderr

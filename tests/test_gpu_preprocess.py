"""preprocess_image on the GPU against oracle/preprocess_oracle.py: bit-exact,
because both use the same f32 operation order; and against the reference's PNG fixtures approximately
(the JPEG decoder differs, see tests/test_oracle_preprocess.py)."""
import os

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import text_detection as td
from ocr_rs_amd import weights as W
from oracle import preprocess_oracle as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    n = td.resnet18(W.pack_blob(W.make_det_weights(0)), 0)
    yield n
    n.close()


@pytest.mark.parametrize("name,resized", [("img224", (600, 800)), ("img55", (800, 533)), ("img494", (800, 800)),
                                          ("img545", (537, 800))])
def test_preprocess_reference_images(net, golden_dir, name, resized):
    rgba = np.array(Image.open(os.path.join(golden_dir, "text_det", name + ".jpg")).convert("RGBA"))
    gray, ax, ay = td.preprocess_image(net, rgba, (800, 800))
    ogray, oax, oay = P.preprocess_image(rgba, 800, 800)
    assert (ax, ay) == (oax, oay)
    assert np.array_equal(gray, ogray)
    exp = np.array(Image.open(os.path.join(golden_dir, f"preprocessed_{name}.png")).convert("L"))
    d = np.abs(gray.astype(int) - exp.astype(int))[:resized[1], :resized[0]]
    assert (d == 0).mean() >= 0.94 and d.max() <= 2


@pytest.mark.parametrize("w,h,tw,th", [(1600, 1200, 640, 640), (97, 211, 640, 640), (640, 640, 640, 640), (3, 5, 64, 32)])
def test_preprocess_synthetic_shapes(net, w, h, tw, th):
    """Down-scaling (support > 1), odd sizes, identity and tiny inputs; also the f32 frame output."""
    rng = np.random.RandomState(w * 7 + h)
    rgba = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    gray, f32, ax, ay = net.handle.preprocess_image(rgba, tw, th, want_f32=True)
    ogray, oax, oay = P.preprocess_image(rgba, tw, th)
    assert (ax, ay) == (oax, oay)
    assert np.array_equal(gray, ogray)
    assert np.array_equal(f32[0, 0], gray.astype(np.float32))     # raw 0..255, no normalisation (mod.rs:46-54)


def test_preprocess_then_detect_pipeline(net, golden_dir):
    """run_text_detection's front half on the GPU: preprocess -> forward_t -> polygons (mod.rs:23-67)."""
    rgba = np.array(Image.open(os.path.join(golden_dir, "text_det", "img55.jpg")).convert("RGBA"))
    gray, f32, ax, ay = net.handle.preprocess_image(rgba, 800, 800, want_f32=True)
    pred = net.forward_t(f32)
    assert pred.shape == (1, 1, 800, 800) and np.isfinite(pred).all()
    res = td.get_boxes_and_box_scores(net, pred, np.array([[ax, ay]]), skip_degenerate=True)
    assert len(res.polygons) == 1

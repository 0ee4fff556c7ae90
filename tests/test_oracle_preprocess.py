"""oracle/preprocess_oracle.py against the reference's preprocess fixtures (image_ops.rs:805-1008).
The reference decoded the JPEGs with jpeg-decoder 0.1.20; PIL's libjpeg differs in the last bit, so
the PNG fixtures are matched approximately; sizes and adjust values are exact."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import preprocess_oracle as P

CASES = [  # name, original (w,h), resized (w,h): image_ops.rs:902-916 and the test/ pair
    ("img224", (180, 240), (600, 800)),
    ("img55", (300, 200), (800, 533)),
    ("img494", (200, 200), (800, 800)),
    ("img545", (184, 274), (537, 800)),
]


@pytest.mark.parametrize("name,orig,resized", CASES)
def test_preprocess_oracle_vs_reference_fixture(golden_dir, name, orig, resized):
    rgba = np.array(Image.open(os.path.join(golden_dir, "text_det", name + ".jpg")).convert("RGBA"))
    assert rgba.shape[:2] == (orig[1], orig[0])
    out, ax, ay = P.preprocess_image(rgba, 800, 800)
    assert (ax, ay) == (resized[0] / orig[0], resized[1] / orig[1])          # exact f64 ratios
    exp = np.array(Image.open(os.path.join(golden_dir, f"preprocessed_{name}.png")).convert("L"))
    assert out.shape == exp.shape == (800, 800)
    assert (out[resized[1]:, :] == 0).all() and (out[:, resized[0]:] == 0).all()   # zero padding
    d = np.abs(out.astype(int) - exp.astype(int))[:resized[1], :resized[0]]
    assert (d == 0).mean() >= 0.94 and d.max() <= 2


def test_resize_dimensions_matches_reference_adjust_values():
    assert P.resize_dimensions(180, 240, 800, 800) == (600, 800)
    assert P.resize_dimensions(300, 200, 800, 800) == (800, 533)
    assert P.resize_dimensions(184, 274, 800, 800) == (537, 800)
    assert P.resize_dimensions(3000, 10, 640, 640) == (640, 2)
    assert P.resize_dimensions(10, 3000, 640, 640) == (2, 640)

"""Pins oracle/postproc_oracle.py against every post-processing known-answer
test of the reference (metrics.rs:406-646)."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import postproc_oracle as O
from tests import kat_postproc as K


def test_min_area_bounding_box_kat():
    box, sside = O.get_min_area_bounding_box(K.MIN_AREA_BOX_IN)
    assert box == K.MIN_AREA_BOX_OUT
    assert abs(sside - K.MIN_AREA_BOX_SSIDE) < np.finfo(np.float64).eps


@pytest.mark.parametrize("pts,expected", K.BOX_SCORE_CASES)
def test_box_score_kat(pts, expected):
    pred = np.array(K.BOX_SCORE_MAP, dtype=np.float64).reshape(5, 5)
    assert O.box_score_fast(pred, pts) == expected


def test_binarize_kat():
    pred = np.array(K.BINARIZE_IN, dtype=np.float64).reshape(5, 5)
    out = O.binarize(pred, K.BINARIZE_THRESH)
    assert out.dtype == np.uint8
    assert out.reshape(-1).tolist() == K.BINARIZE_OUT


def _img55(golden_dir):
    img = np.array(Image.open(os.path.join(golden_dir, "gt_shrinked_img55.png")).convert("L"))
    bitmap = (img.astype(np.float64) / 255.0).astype(np.uint8)
    pred = img.astype(np.float64) / 255.0
    return pred, bitmap


@pytest.mark.parametrize("adj,polys", [((1.0, 1.0), K.IMG55_POLYS_ADJ1), ((2.0, 2.0), K.IMG55_POLYS_ADJ2)])
def test_get_polygons_from_bitmap_kat(golden_dir, adj, polys):
    pred, bitmap = _img55(golden_dir)
    boxes, scores = O.get_polygons_from_bitmap(pred, bitmap, adj)
    assert boxes == polys
    assert scores == K.IMG55_SCORES


def test_contour_lengths_img55(golden_dir):
    # four outer borders of 239/463/196/505 points
    _, bitmap = _img55(golden_dir)
    cs = O.find_contours(bitmap * 255)
    assert [len(c) for c in cs] == [239, 463, 196, 505]


# ---- the compiled CPU baseline (oracle/postproc_cpu.cpp, what bench.py's cpu_baseline.postprocess times) is held to the
# same known answers and to the Python restatement
@pytest.fixture(scope="module")
def cpu_lib():
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-s", "-C", here])
    from oracle import postproc_cpu as PC
    return PC


@pytest.mark.parametrize("adj,polys", [((1.0, 1.0), K.IMG55_POLYS_ADJ1), ((2.0, 2.0), K.IMG55_POLYS_ADJ2)])
def test_compiled_cpu_postprocess_reproduces_the_reference_kat(golden_dir, cpu_lib, adj, polys):
    img = np.array(Image.open(os.path.join(golden_dir, "gt_shrinked_img55.png")).convert("L"))
    pred = (img.astype(np.float64) / 255.0).astype(np.float32).reshape(1, 1, 800, 800)
    for threads in (1, 4):
        got_p, got_s = cpu_lib.get_boxes_and_box_scores(pred, np.array([adj]), threads=threads)
        assert got_p[0] == polys and got_s[0] == K.IMG55_SCORES


def test_compiled_cpu_postprocess_matches_python_oracle_on_real_valued_maps(golden_dir, cpu_lib):
    """Maps with fractional probabilities (scores are real f64 means, not 1.0), several images, several threads."""
    rng = np.random.RandomState(3)
    names = ["gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]
    maps = []
    for nm in names:
        g = np.array(Image.open(os.path.join(golden_dir, nm)).convert("L"))[80:720, 80:720]
        maps.append(np.where(g > 127, 0.75 + 0.25 * rng.rand(640, 640), 0.1 * rng.rand(640, 640)).astype(np.float32))
    pred = np.stack(maps)[:, None]
    adj = np.array([[1.0, 1.0], [0.8, 0.9], [1.25, 0.5]])
    want_p, want_s = O.get_boxes_and_box_scores(pred, adj)
    got_p, got_s = cpu_lib.get_boxes_and_box_scores(pred, adj, threads=3)
    assert got_p == want_p and sum(len(p) for p in want_p) > 3
    for a, b in zip(got_s, want_s):
        assert np.allclose(a, b, rtol=0, atol=1e-12)
    assert cpu_lib.get_boxes_and_box_scores(pred, adj, threads=2, counts_only=True)[0] == sum(len(p) for p in want_p)


def test_oracle_hypot_is_libm_hypot():
    """The reference sums a polygon's perimeter and measures the box sides with Rust's f64::hypot = libm's
    (/root/reference/src/polygon.rs:27, src/text_detection/metrics.rs:145-146, geo 0.15 euclidean_length).  The oracle's
    restatement (O.hypot_libm) against libm.so.6 itself on every integer pair 0 <= b <= a <= 4100 (8.4 M pairs: the range of
    edge vectors an 800 x 800 .. 4096 x 4096 map can hold), and the evidence that the distinction matters: CPython's math.hypot
    and sqrt(a^2 + b^2) each differ from libm on some of them."""
    import ctypes as C
    import math
    libm = C.CDLL("libm.so.6")
    libm.hypot.restype = C.c_double
    libm.hypot.argtypes = [C.c_double, C.c_double]
    lh, oh = libm.hypot, O.hypot_libm
    bad = n = 0
    for a in range(0, 4101):
        fa = float(a)
        for b in range(0, a + 1):
            n += 1
            if oh(fa, float(b)) != lh(fa, float(b)):
                bad += 1
    assert n == 4101 * 4102 // 2 and bad == 0
    # argument order / signs (the kernel orders by magnitude itself)
    for a, b in ((3, -4), (-4, 3), (-5, -12), (0, -7), (-7, 0), (1234, -4099)):
        assert oh(float(a), float(b)) == lh(float(a), float(b)) == oh(float(b), float(a))
    assert any(lh(float(a), float(b)) != math.sqrt(a * a + b * b) for a in range(1, 300) for b in range(1, a))
    assert any(lh(float(a), float(b)) != math.hypot(float(a), float(b)) for a in range(1, 600) for b in range(1, a))

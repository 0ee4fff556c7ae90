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

"""Detection quality metrics (8f row 4): the reference's KATs metrics.rs:648-901 against both the oracle and
the product's host C++ (through the C ABI; no GPU needed)."""
import random

import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import metrics as M
from oracle import metrics_oracle as MO

GT = [[(0, 0), (10, 0), (10, 10), (0, 10)], [(20, 20), (30, 20), (30, 30), (20, 30)]]
P1 = [(1, 1), (10, 0), (10, 10), (0, 10)]


def _both(gt, ign, pred):
    a = M.evaluate_image(gt, ign, pred)
    b = MO.evaluate_image(gt, ign, pred)
    assert (a.precision, a.recall, a.hmean, a.gt_care, a.det_care, a.det_matched) == \
           (b["precision"], b["recall"], b["hmean"], b["gt_care"], b["det_care"], b["det_matched"])
    return a


def test_evaluate_image_test_one_matching_polygon():            # metrics.rs:648-678
    m = _both(GT, [False, False], [P1])
    assert (m.gt_care, m.det_care, m.det_matched) == (2, 1, 1)
    assert m.precision == 1.0 and m.recall == 0.5 and abs(m.hmean - 0.6666666666666666) < 2.3e-16


def test_evaluate_image_test_with_ignored_polygons():           # metrics.rs:680-710
    m = _both(GT, [True, True], [P1])
    assert (m.gt_care, m.det_care, m.det_matched) == (0, 0, 0)
    assert (m.precision, m.recall, m.hmean) == (1.0, 1.0, 1.0)


def test_evaluate_image_test_with_both_matched_polygons():      # metrics.rs:712-748
    m = _both(GT, [False, False], [P1, GT[1]])
    assert (m.gt_care, m.det_care, m.det_matched) == (2, 2, 2)
    assert (m.precision, m.recall, m.hmean) == (1.0, 1.0, 1.0)


def test_validate_measure_test():                                # metrics.rs:750-812
    pred = [[P1], [[(45, 61), (47, 41), (60, 60), (39, 48)]]]
    ms = M.validate_measure([GT, GT], [[False, False], [False, False]], pred, [[0.9], [0.9]])
    assert len(ms) == 2
    assert (ms[0].gt_care, ms[0].det_care, ms[0].det_matched) == (2, 1, 1)
    assert ms[0].precision == 1.0 and ms[0].recall == 0.5 and abs(ms[0].hmean - 0.6666666666666666) < 2.3e-16
    assert (ms[1].gt_care, ms[1].det_care, ms[1].det_matched) == (2, 1, 0)
    assert ms[1].precision == 0.0 and ms[1].recall == 0.0 and ms[1].hmean == 0.0
    # predictions under the 0.6 score threshold are dropped before evaluation (metrics.rs:197-207)
    ms = M.validate_measure([GT], [[False, False]], [[P1]], [[0.59]])
    assert (ms[0].det_care, ms[0].det_matched) == (0, 0)


ITEMS = [M.MetricsItem(1.0, 0.5, 0.6666666666666666, 2, 1, 1), M.MetricsItem(1.0, 1.0, 1.0, 0, 0, 0),
         M.MetricsItem(1.0, 1.0, 1.0, 2, 2, 2), M.MetricsItem(0.3333333333333333, 0.2, 0.25, 5, 3, 1)]


def test_combine_results_test():                                 # metrics.rs:814-853
    assert M.combine_results(ITEMS) == (0.6666666666666666, 0.4444444444444444, 0.5333333333333333)
    assert MO.combine_results([i.__dict__ for i in ITEMS]) == (0.6666666666666666, 0.4444444444444444, 0.5333333333333333)


def test_gather_measure_test():                                  # metrics.rs:855-901
    assert M.gather_measure([ITEMS[:2], ITEMS[2:]]) == (0.6666666666666666, 0.4444444444444444, 0.5333333333333333)


def test_iou_areas_match_oracle_on_random_polygons():
    from ocr_rs_amd import capi
    rnd = random.Random(3)
    import math
    for _ in range(40):
        polys = []
        for _k in range(2):
            k = rnd.randint(3, 9)
            cx, cy = rnd.randint(40, 80), rnd.randint(40, 80)
            polys.append([(int(cx + rnd.uniform(8, 40) * math.cos(2 * math.pi * i / k)),
                           int(cy + rnd.uniform(8, 40) * math.sin(2 * math.pi * i / k))) for i in range(k)])
        if any(len(set(p)) < len(p) for p in polys):
            continue
        a = M.evaluate_image([polys[0]], [False], [polys[1]])
        b = MO.evaluate_image([polys[0]], [False], [polys[1]])
        assert a.det_matched == b["det_matched"], polys

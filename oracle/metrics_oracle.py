"""ORACLE (test infrastructure): detection quality metrics restated in Python.

    evaluate_image   /root/reference/src/text_detection/metrics.rs:255-380
    combine_results  metrics.rs:229-253, validate_measure :191-219, gather_measure :221-227
Polygon intersection / union areas: the reference calls geo-clipper (Clipper 6.x, not vendored); here
they are exact rational areas from Sutherland-style half-plane clipping of triangle fans (independent
of the product's arrangement code).  Pinned by the reference KATs metrics.rs:648-901
(tests/test_metrics.py).
"""
from __future__ import annotations

from fractions import Fraction
from typing import List, Sequence, Tuple

Poly = Sequence[Tuple[int, int]]


def _area2(p) -> Fraction:
    s = Fraction(0)
    n = len(p)
    for i in range(n):
        s += p[i][0] * p[(i + 1) % n][1] - p[(i + 1) % n][0] * p[i][1]
    return s


def _clip_convex(subject, a, b, c):
    """Clip polygon `subject` (list of rational points) to the CCW triangle a,b,c."""
    out = list(subject)
    for p, q in ((a, b), (b, c), (c, a)):
        if not out:
            break
        res = []

        def side(pt):
            return (q[0] - p[0]) * (pt[1] - p[1]) - (q[1] - p[1]) * (pt[0] - p[0])
        for i in range(len(out)):
            cur, nxt = out[i], out[(i + 1) % len(out)]
            sc, sn = side(cur), side(nxt)
            if sc >= 0:
                res.append(cur)
            if (sc > 0 and sn < 0) or (sc < 0 and sn > 0):
                t = Fraction(sc, sc - sn)
                res.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
        out = res
    return out


def _triangles(poly):
    """Signed triangle fan of a simple polygon (works for concave ones through signed areas)."""
    p0 = poly[0]
    for i in range(1, len(poly) - 1):
        yield p0, poly[i], poly[i + 1]


def intersection_area(p1: Poly, p2: Poly) -> float:
    a = [(Fraction(x), Fraction(y)) for x, y in p1]
    b = [(Fraction(x), Fraction(y)) for x, y in p2]
    total = Fraction(0)
    for ta in _triangles(a):
        sa = _area2(ta)
        if sa == 0:
            continue
        ta_ccw = ta if sa > 0 else (ta[0], ta[2], ta[1])
        for tb in _triangles(b):
            sb = _area2(tb)
            if sb == 0:
                continue
            tb_ccw = tb if sb > 0 else (tb[0], tb[2], tb[1])
            inter = _clip_convex(list(ta_ccw), *tb_ccw)
            if len(inter) >= 3:
                total += abs(_area2(inter)) * (1 if sa > 0 else -1) * (1 if sb > 0 else -1)
    return float(abs(total) / 2)


def polygon_area(p: Poly) -> float:
    return float(abs(_area2([(Fraction(x), Fraction(y)) for x, y in p])) / 2)


def union_area(p1: Poly, p2: Poly) -> float:
    return polygon_area(p1) + polygon_area(p2) - intersection_area(p1, p2)


def evaluate_image(gt: Sequence[Poly], ignore_flags: Sequence[bool], pred: Sequence[Poly]):
    gt_dc = [n for n in range(len(gt)) if ignore_flags[n]]
    det_dc = []
    for d, pd in enumerate(pred):
        for g in gt_dc:
            inter = intersection_area(gt[g], pd)
            area = polygon_area(pd)
            if (0.0 if area == 0 else inter / area) > 0.5:
                det_dc.append(d)
                break
    matched = 0
    if gt and pred:
        gu, du = [0] * len(gt), [0] * len(pred)
        for g in range(len(gt)):
            for d in range(len(pred)):
                iou = intersection_area(pred[d], gt[g]) / union_area(pred[d], gt[g])
                # the reference tests the DETECTION index against the gt don't-care list (metrics.rs:331)
                if gu[g] == 0 and du[d] == 0 and g not in gt_dc and d not in gt_dc and iou > 0.5:
                    gu[g] = du[d] = 1
                    matched += 1
    gt_care, det_care = len(gt) - len(gt_dc), len(pred) - len(det_dc)
    if gt_care == 0:
        recall, precision = 1.0, (0.0 if det_care > 0 else 1.0)
    else:
        recall = matched / gt_care
        precision = 0.0 if det_care == 0 else matched / det_care
    hmean = 0.0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)
    return dict(precision=precision, recall=recall, hmean=hmean, gt_care=gt_care, det_care=det_care, det_matched=matched)


def combine_results(items):
    gt = sum(i["gt_care"] for i in items)
    det = sum(i["det_care"] for i in items)
    m = sum(i["det_matched"] for i in items)
    r = m / gt if gt else 0.0
    p = m / det if det else 0.0
    h = 2 * (r * p) / (r + p) if r + p else 0.0
    return p, r, h

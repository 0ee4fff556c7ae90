"""Host-side mirror of the reference's detection quality metrics over the C ABI (no GPU involved).

    validate_measure(polygons, ignore_tags, pred, scores)   text_detection/metrics.rs:191-219
    gather_measure(metrics) / combine_results(results)      metrics.rs:221-253
    evaluate_image(gt_points, ignore_flags, pred)           metrics.rs:255-380
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import capi

Poly = Sequence[Tuple[int, int]]


@dataclass
class MetricsItem:                      # metrics.rs:22-30
    precision: float
    recall: float
    hmean: float
    gt_care: int
    det_care: int
    det_matched: int


def _csr(polys: Sequence[Poly]):
    xy = np.asarray([c for p in polys for v in p for c in v], dtype=np.uint32)
    off = np.zeros(len(polys) + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in polys])
    return xy, off


def evaluate_image(gt_points: Sequence[Poly], ignore_flags: Sequence[bool], pred: Sequence[Poly]) -> MetricsItem:
    gxy, goff = _csr(gt_points)
    pxy, poff = _csr(pred)
    ign = np.asarray(ignore_flags, dtype=np.uint8)
    out = capi.MetricsItem()
    capi.check(capi.lib().ocr_evaluate_image(capi._ptr(gxy) if gxy.size else None, capi._ptr(goff), len(gt_points),
                                             capi._ptr(ign) if ign.size else None, capi._ptr(pxy) if pxy.size else None,
                                             capi._ptr(poff), len(pred), C.byref(out)))
    return MetricsItem(out.precision, out.recall, out.hmean, out.gt_care, out.det_care, out.det_matched)


def validate_measure(polygons, ignore_tags, pred, scores) -> List[MetricsItem]:
    box_thresh = 0.6                                                   # metrics.rs:197
    result = []
    for cur_polys, cur_ign, cur_pred, cur_scores in zip(polygons, ignore_tags, pred, scores):
        kept = [p for s, p in zip(cur_scores, cur_pred) if s >= box_thresh]
        result.append(evaluate_image(cur_polys, cur_ign, kept))
    return result


def combine_results(results: Sequence[MetricsItem]) -> Tuple[float, float, float]:
    arr = (capi.MetricsItem * len(results))(*[capi.MetricsItem(r.precision, r.recall, r.hmean, r.gt_care, r.det_care,
                                                               r.det_matched) for r in results])
    p, r, h = C.c_double(), C.c_double(), C.c_double()
    capi.check(capi.lib().ocr_combine_results(arr, len(results), C.byref(p), C.byref(r), C.byref(h)))
    return p.value, r.value, h.value


def gather_measure(metrics: Sequence[Sequence[MetricsItem]]) -> Tuple[float, float, float]:
    return combine_results([m for batch in metrics for m in batch])

"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in plain Python, of the reference's detection post-processing

    get_boxes_and_box_scores     /root/reference/src/text_detection/metrics.rs:37-56
    get_polygons_from_bitmap     metrics.rs:58-127
    binarize                     metrics.rs:129-131
    get_min_area_bounding_box    metrics.rs:133-148
    box_score_fast               metrics.rs:150-184
    expand_polygon/clip_polygon  /root/reference/src/polygon.rs:13-56

The arithmetic of those functions lives in crates that are NOT vendored in
/root/reference:
    imageproc 0.22.0  (find_contours, arc_length, approximate_polygon_dp,
                       min_area_rect, draw_polygon_mut)      Cargo.lock:626-627
    geo 0.15.0        (unsigned_area, euclidean_length)       Cargo.lock:485-486
    geo-clipper 0.4.1-alpha.0 @54577fb / clipper-sys 0.3.3-alpha.0 @7f6d3a0
                      (Clipper 6.x ClipperOffset + union)     Cargo.lock:242-244,499-501
Their published algorithms are restated below and PINNED against every
known-answer test the reference holds for this path (metrics.rs:406-646); see
tests/test_oracle_postproc.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Known unpinned corner: Clipper's union is restated as "outer boundary of the
positive-winding region of the raw offset ring, emitted in Clipper's
BuildResult order" (see clipper_offset_expand). The reference KATs pin the
convex and the concave-loop cases; degenerate touching/collinear outputs are
not pinned by any reference test.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import List, Sequence, Tuple

import numpy as np

Pt = Tuple[int, int]


# --------------------------------------------------------------------------
# binarize  (metrics.rs:129-131)
# --------------------------------------------------------------------------
def binarize(pred: np.ndarray, thresh: float = 0.6) -> np.ndarray:
    """`pred.gt(thresh).to_kind(Uint8)`.

    libtorch compares in the tensor's dtype: an f32 tensor against a wrapped
    scalar compares with (float)thresh; an f64 tensor (the reference KAT,
    metrics.rs:486-508) compares in f64.  Strict '>'.
    """
    if pred.dtype == np.float32:
        return (pred > np.float32(thresh)).astype(np.uint8)
    return (pred > thresh).astype(np.uint8)


# --------------------------------------------------------------------------
# imageproc::contours::find_contours  (called at metrics.rs:78)
# --------------------------------------------------------------------------
_DIRS = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]  # W NW N NE E SE S SW


def find_contours(img: np.ndarray) -> List[List[Pt]]:
    """Suzuki-Abe border following as published in imageproc 0.22.0
    (`find_contours_with_threshold(image, 0)`): every border (outer and hole)
    is returned in discovery order; the reference keeps all of them
    (metrics.rs:78-81 maps `c.points` without filtering border_type).
    """
    h, w = img.shape
    v = (img > 0).astype(np.int64)
    vals = v.tolist()  # row-major python lists: faster scalar access
    contours: List[List[Pt]] = []
    curr_border = 1

    def nonzero(x: int, y: int) -> bool:
        return 0 <= x < w and 0 <= y < h and vals[y][x] != 0

    for y in range(h):
        row = vals[y]
        for x in range(w):
            if row[x] == 0:
                continue
            adj = None
            if row[x] == 1 and x > 0 and row[x - 1] == 0:
                adj = (x - 1, y)
            elif row[x] > 0 and x + 1 < w and row[x + 1] == 0:
                adj = (x + 1, y)
            if adj is None:
                continue
            curr_border += 1
            pts: List[Pt] = []
            cx, cy = x, y
            start = _DIRS.index((adj[0] - cx, adj[1] - cy))
            # clockwise search (W,NW,N,... order) starting at the adjacent zero pixel
            pos1 = None
            for k in range(8):
                d = _DIRS[(start + k) % 8]
                if nonzero(cx + d[0], cy + d[1]):
                    pos1 = (cx + d[0], cy + d[1])
                    break
            if pos1 is None:
                pts.append((x, y))
                row[x] = -curr_border
            else:
                pos2 = pos1
                pos3 = (cx, cy)
                while True:
                    pts.append(pos3)
                    base = _DIRS.index((pos2[0] - pos3[0], pos2[1] - pos3[1]))
                    # counter-clockwise: deque rotated so that front == dir(pos2);
                    # iter().rev() visits base-1, base-2, ..., base
                    pos4 = None
                    is_right_edge = False
                    for k in range(1, 9):
                        d = _DIRS[(base - k) % 8]
                        if nonzero(pos3[0] + d[0], pos3[1] + d[1]):
                            pos4 = (pos3[0] + d[0], pos3[1] + d[1])
                            break
                        if d == (1, 0):
                            is_right_edge = True
                    assert pos4 is not None
                    if pos3[0] + 1 == w or is_right_edge:
                        vals[pos3[1]][pos3[0]] = -curr_border
                    elif vals[pos3[1]][pos3[0]] == 1:
                        vals[pos3[1]][pos3[0]] = curr_border
                    if pos4 == (cx, cy) and pos3 == pos1:
                        break
                    pos2 = pos3
                    pos3 = pos4
            contours.append(pts)
    return contours


# --------------------------------------------------------------------------
# imageproc::geometry::{arc_length, approximate_polygon_dp}  (metrics.rs:87-91)
# --------------------------------------------------------------------------
def _dist(a: Pt, b: Pt) -> float:
    dx = float(a[0]) - float(b[0])
    dy = float(a[1]) - float(b[1])
    return math.sqrt(dx * dx + dy * dy)


def arc_length(pts: Sequence[Pt], closed: bool) -> float:
    length = 0.0
    for i in range(len(pts) - 1):
        length += _dist(pts[i], pts[i + 1])
    if len(pts) > 2 and closed:
        length += _dist(pts[0], pts[-1])
    return length


def approximate_polygon_dp(curve: Sequence[Pt], epsilon: float, closed: bool) -> List[Pt]:
    end = len(curve) - 1
    x0, y0 = float(curve[0][0]), float(curve[0][1])
    x1, y1 = float(curve[end][0]), float(curve[end][1])
    a = y0 - y1
    b = x1 - x0
    c = x0 * y1 - x1 * y0
    norm = math.sqrt(a * a + b * b)
    dmax = 0.0
    index = 0
    for i in range(1, end + 1):
        num = abs(a * float(curve[i][0]) + b * float(curve[i][1]) + c)
        d = num / norm if norm != 0.0 else float("nan")
        if d > dmax:
            index = i
            dmax = d
    if dmax > epsilon:
        p1 = approximate_polygon_dp(curve[: index + 1], epsilon, False)
        p2 = approximate_polygon_dp(curve[index:], epsilon, False)
        p1.pop()
        res = p1 + p2
    else:
        res = [curve[0], curve[end]]
    if closed:
        res.pop()
    return res


# --------------------------------------------------------------------------
# imageproc::drawing::draw_polygon_mut + box_score_fast  (metrics.rs:150-184)
# --------------------------------------------------------------------------
def _round_half_away_f32(v: np.float32) -> int:
    f = float(v)
    return int(math.floor(f + 0.5)) if f >= 0 else -int(math.floor(-f + 0.5))


def _bresenham(canvas: np.ndarray, s: Pt, e: Pt) -> None:
    hh, ww = canvas.shape
    x0, y0, x1, y1 = float(s[0]), float(s[1]), float(e[0]), float(e[1])
    steep = abs(y1 - y0) > abs(x1 - x0)
    if steep:
        x0, y0, x1, y1 = y0, x0, y1, x1
    if x0 > x1:
        x0, x1, y0, y1 = x1, x0, y1, y0
    dx = np.float32(x1 - x0)
    dy = np.float32(abs(y1 - y0))
    err = np.float32(dx / np.float32(2))
    ystep = 1 if y0 < y1 else -1
    x, y, endx = int(x0), int(y0), int(x1)
    while x <= endx:
        px, py = (y, x) if steep else (x, y)
        if 0 <= px < ww and 0 <= py < hh:
            canvas[py, px] = 255
        x += 1
        err = np.float32(err - dy)
        if err < 0:
            y += ystep
            err = np.float32(err + dx)


def draw_polygon(canvas: np.ndarray, poly: Sequence[Pt]) -> None:
    """imageproc 0.22.0 draw_polygon_mut: scanline fill then Bresenham outline."""
    if not poly:
        return
    if poly[0] == poly[-1]:
        raise ValueError("First point == last point")  # imageproc panics
    hh, ww = canvas.shape
    ys = [p[1] for p in poly]
    y_min = max(0, min(min(ys), hh - 1))
    y_max = max(0, min(max(ys), hh - 1))
    closed = list(poly) + [poly[0]]
    for y in range(y_min, y_max + 1):
        inter: List[int] = []
        for i in range(len(closed) - 1):
            p0, p1 = closed[i], closed[i + 1]
            if (p0[1] <= y <= p1[1]) or (p1[1] <= y <= p0[1]):
                if p0[1] == p1[1]:
                    inter.append(p0[0])
                    inter.append(p1[0])
                elif p0[1] == y or p1[1] == y:
                    if p1[1] > y:
                        inter.append(p0[0])
                    if p0[1] > y:
                        inter.append(p1[0])
                else:
                    frac = np.float32(y - p0[1]) / np.float32(p1[1] - p0[1])
                    v = np.float32(p0[0]) + np.float32(frac * np.float32(p1[0] - p0[0]))
                    inter.append(_round_half_away_f32(v))
        inter.sort()
        for k in range(0, len(inter) - 1, 2):
            frm = min(inter[k], ww)
            to = min(inter[k + 1], ww - 1)
            if frm < ww and to >= 0:
                frm = max(0, frm)
                to = max(0, to)
                if to >= frm:
                    canvas[y, frm:to + 1] = 255
    for i in range(len(closed) - 1):
        _bresenham(canvas, closed[i], closed[i + 1])


def box_score_fast(pred: np.ndarray, points: Sequence[Pt]) -> float:
    """metrics.rs:150-184, including its naming quirk: size[-2] is called `w`
    and clamps x, size[-1] is called `h` and clamps y."""
    w = pred.shape[-2]
    h = pred.shape[-1]
    xs = [p[0] for p in points]
    ys = [p[1] for p in points]
    min_x = min(max(min(xs), 0), w - 1)
    max_x = min(max(max(xs), 0), w - 1)
    min_y = min(max(min(ys), 0), h - 1)
    max_y = min(max(max(ys), 0), h - 1)
    mask = np.zeros((max_y - min_y + 1, max_x - min_x + 1), dtype=np.uint8)
    moved = [(p[0] - min_x, p[1] - min_y) for p in points]
    draw_polygon(mask, moved)
    m = (mask // 255).astype(np.uint8)
    part = pred[min_y:max_y + 1, min_x:max_x + 1]
    # f32 * u8 -> f32 (exact: mask is 0/1), summed in f64
    s = np.sum((part * m).astype(np.float64))
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(s / np.sum(m, dtype=np.float64))   # empty mask: 0/0 = NaN, like tch (no error)


def polygon_mask_count(points: Sequence[Pt], shape: Tuple[int, int]) -> int:
    xs = [p[0] for p in points]
    ys = [p[1] for p in points]
    mask = np.zeros((max(ys) - min(ys) + 1, max(xs) - min(xs) + 1), dtype=np.uint8)
    draw_polygon(mask, [(p[0] - min(xs), p[1] - min(ys)) for p in points])
    return int(mask.astype(bool).sum())


# --------------------------------------------------------------------------
# imageproc::geometry::min_area_rect + get_min_area_bounding_box (metrics.rs:133-148)
# --------------------------------------------------------------------------
def _orient(p: Pt, q: Pt, r: Pt) -> int:
    val = (q[1] - p[1]) * (r[0] - q[0]) - (q[0] - p[0]) * (r[1] - q[1])
    return 0 if val == 0 else (1 if val > 0 else 2)  # 1 clockwise, 2 counter-clockwise


def convex_hull(points: Sequence[Pt]) -> List[Pt]:
    import functools
    if not points:
        return []
    pts = list(points)
    sp = 0
    for i in range(1, len(pts)):
        if pts[i][1] < pts[sp][1] or (pts[i][1] == pts[sp][1] and pts[i][0] < pts[sp][0]):
            sp = i
    start = pts[sp]
    pts[0], pts[sp] = pts[sp], pts[0]
    pts = pts[1:]

    def cmp(a: Pt, b: Pt) -> int:
        o = _orient(start, a, b)
        if o == 0:
            return -1 if _dist(start, a) < _dist(start, b) else 1
        return 1 if o == 1 else -1

    pts.sort(key=functools.cmp_to_key(cmp))
    rem: List[Pt] = []
    i = 0
    while i < len(pts):
        p = pts[i]
        while i + 1 < len(pts) and _orient(start, p, pts[i + 1]) == 0:
            i += 1
            p = pts[i]
        rem.append(p)
        i += 1
    stack = [start]
    for p in rem:
        while len(stack) > 1 and _orient(stack[-2], stack[-1], p) != 2:
            stack.pop()
        stack.append(p)
    return stack


def min_area_rect(points: Sequence[Pt]) -> List[Pt]:
    hull = convex_hull(points)
    if len(hull) == 1:
        return [hull[0]] * 4
    if len(hull) == 2:
        return [hull[0], hull[1], hull[1], hull[0]]
    angles: List[float] = []
    for i in range(len(hull) - 1):
        ex = float(hull[i + 1][0]) - float(hull[i][0])
        ey = float(hull[i + 1][1]) - float(hull[i][1])
        ang = abs(math.fmod(math.atan2(ey, ex) + math.pi, math.pi / 2.0))
        if not angles or angles[-1] != ang:
            angles.append(ang)
    min_area = float("inf")
    res = [(0.0, 0.0)] * 4
    for ang in angles:
        s, c = math.sin(ang), math.cos(ang)
        rot = [(p[0] * c + p[1] * s, p[1] * c - p[0] * s) for p in hull]  # rotate by -angle
        min_x = min(p[0] for p in rot)
        max_x = max(p[0] for p in rot)
        min_y = min(p[1] for p in rot)
        max_y = max(p[1] for p in rot)
        area = (max_x - min_x) * (max_y - min_y)
        if area < min_area:
            min_area = area

            def back(px: float, py: float) -> Tuple[float, float]:  # rotate by +angle
                return (px * c - py * s, py * c + px * s)
            res = [back(max_x, min_y), back(min_x, min_y), back(min_x, max_y), back(max_x, max_y)]
    res.sort(key=lambda p: p[0])
    i1 = 0 if res[1][1] > res[0][1] else 1
    i2 = 2 if res[3][1] > res[2][1] else 3
    i3 = 3 if res[3][1] > res[2][1] else 2
    i4 = 1 if res[1][1] > res[0][1] else 0
    return [
        (int(math.floor(res[i1][0])), int(math.floor(res[i1][1]))),
        (int(math.ceil(res[i2][0])), int(math.floor(res[i2][1]))),
        (int(math.ceil(res[i3][0])), int(math.ceil(res[i3][1]))),
        (int(math.floor(res[i4][0])), int(math.ceil(res[i4][1]))),
    ]


def hypot_libm(x: float, y: float) -> float:
    """libm's hypot as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/e_hypot.c, the branch without FMA that the x86-64 build
    runs) for operands that need no scaling.  Rust's f64::hypot - what geo 0.15's euclidean_length calls
    (/root/reference/src/polygon.rs:27, src/text_detection/metrics.rs:145-146) - is this routine, NOT CPython's math.hypot
    (its own correctly rounded algorithm) and NOT sqrt(dx*dx + dy*dy): each differs from libm by an ulp on ~0.6 % of integer
    pairs.  Every operation below is one separately rounded IEEE double operation, which is what Python floats give, so the
    restatement reproduces libm bit for bit (tests/test_oracle_postproc.py checks it against libm.so.6 on 8.4 M integer pairs)."""
    ax, ay = abs(x), abs(y)
    if ax < ay:
        ax, ay = ay, ax
    if ay == 0.0:
        return ax
    h = math.sqrt(ax * ax + ay * ay)
    if h <= 2.0 * ay:
        delta = h - ay
        t1 = ax * (2.0 * delta - ax)
        t2 = (delta - 2.0 * (ax - ay)) * delta
    else:
        delta = h - ax
        t1 = 2.0 * delta * (ax - 2.0 * ay)
        t2 = (4.0 * delta - ay) * ay + delta * delta
    h -= (t1 + t2) / (2.0 * h)
    return h



def get_min_area_bounding_box(contour: Sequence[Pt]) -> Tuple[List[Pt], float]:
    b = sorted(min_area_rect(contour), key=lambda p: p[0])  # stable, by x (metrics.rs:138)
    i1 = 0 if b[1][1] > b[0][1] else 1
    i2 = 2 if b[3][1] > b[2][1] else 3
    i3 = 3 if b[3][1] > b[2][1] else 2
    i4 = 1 if b[1][1] > b[0][1] else 0
    res = [b[i1], b[i2], b[i3], b[i4]]
    wlen = hypot_libm(float(res[0][0]) - float(res[1][0]), float(res[0][1]) - float(res[1][1]))
    hlen = hypot_libm(float(res[0][0]) - float(res[3][0]), float(res[0][1]) - float(res[3][1]))
    return res, min(wlen, hlen)


# --------------------------------------------------------------------------
# polygon.rs:13-56  clip_polygon / expand_polygon  (Clipper offset, Miter 2.0)
# --------------------------------------------------------------------------
def _cround(v: float) -> int:
    """Clipper's Round(): half away from zero via truncation."""
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def _shoelace2(ring: Sequence[Pt]) -> int:
    s = 0
    n = len(ring)
    for i in range(n):
        x0, y0 = ring[i]
        x1, y1 = ring[(i + 1) % n]
        s += x0 * y1 - x1 * y0
    return s


def offset_distance(poly: Sequence[Pt], factor: float) -> float:
    """polygon.rs:27: unsigned_area * factor / closed exterior length (geo 0.15)."""
    n = len(poly)
    area = abs(_shoelace2(poly)) / 2.0
    per = 0.0
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        per += hypot_libm(float(b[0] - a[0]), float(b[1] - a[1]))
    return area * factor / per


def raw_offset_ring(poly: Sequence[Pt], delta: float, miter_limit: float = 2.0) -> List[Pt]:
    """ClipperOffset::DoOffset for one closed polygon, JoinType Miter.
    Returns the raw (possibly self-intersecting) ring handed to the union."""
    # AddPath: strip closing duplicates and consecutive duplicates
    src: List[Pt] = []
    pts = list(poly)
    while len(pts) > 1 and pts[0] == pts[-1]:
        pts.pop()
    for p in pts:
        if not src or src[-1] != p:
            src.append(p)
    n = len(src)
    if n < 3:
        return []
    # FixOrientations: Clipper wants Area >= 0 (shoelace > 0 in numeric coords)
    if _shoelace2(src) < 0:
        src.reverse()
    miter_lim = 2.0 / (miter_limit * miter_limit) if miter_limit > 2.0 else 0.5
    normals = []
    for i in range(n):
        a, b = src[i], src[(i + 1) % n]
        dx, dy = float(b[0] - a[0]), float(b[1] - a[1])
        f = 1.0 / math.sqrt(dx * dx + dy * dy)
        dx *= f
        dy *= f
        normals.append((dy, -dx))
    out: List[Pt] = []
    k = n - 1
    for j in range(n):
        nk, nj = normals[k], normals[j]
        sx, sy = src[j]
        sin_a = nk[0] * nj[1] - nj[0] * nk[1]
        done = False
        if abs(sin_a * delta) < 1.0:
            cos_a = nk[0] * nj[0] + nj[1] * nk[1]
            if cos_a > 0:
                out.append((_cround(sx + nk[0] * delta), _cround(sy + nk[1] * delta)))
                done = True
        elif sin_a > 1.0:
            sin_a = 1.0
        elif sin_a < -1.0:
            sin_a = -1.0
        if not done:
            if sin_a * delta < 0:
                out.append((_cround(sx + nk[0] * delta), _cround(sy + nk[1] * delta)))
                out.append((sx, sy))
                out.append((_cround(sx + nj[0] * delta), _cround(sy + nj[1] * delta)))
            else:
                r = 1.0 + (nj[0] * nk[0] + nj[1] * nk[1])
                if r >= miter_lim:
                    q = delta / r
                    out.append((_cround(sx + (nk[0] + nj[0]) * q), _cround(sy + (nk[1] + nj[1]) * q)))
                else:  # DoSquare
                    dxx = math.tan(math.atan2(sin_a, nk[0] * nj[0] + nk[1] * nj[1]) / 4.0)
                    out.append((_cround(sx + delta * (nk[0] - nk[1] * dxx)),
                                _cround(sy + delta * (nk[1] + nk[0] * dxx))))
                    out.append((_cround(sx + delta * (nj[0] + nj[1] * dxx)),
                                _cround(sy + delta * (nj[1] - nj[0] * dxx))))
        k = j
    return out


def _seg_intersection(p1: Pt, p2: Pt, p3: Pt, p4: Pt):
    """Proper/touching intersection of closed segments p1p2 and p3p4 as exact
    rationals (t on p1p2, u on p3p4, point). None when parallel or disjoint."""
    d1x, d1y = p2[0] - p1[0], p2[1] - p1[1]
    d2x, d2y = p4[0] - p3[0], p4[1] - p3[1]
    den = d1x * d2y - d1y * d2x
    if den == 0:
        return None
    wx, wy = p3[0] - p1[0], p3[1] - p1[1]
    t = Fraction(wx * d2y - wy * d2x, den)
    u = Fraction(wx * d1y - wy * d1x, den)
    if t < 0 or t > 1 or u < 0 or u > 1:
        return None
    return t, u, (p1[0] + t * d1x, p1[1] + t * d1y)


def _winding(ring: Sequence[Pt], qx: Fraction, qy: Fraction) -> int:
    """Winding number of `ring` about the (non-boundary) rational point q;
    +1 for a ring with positive shoelace area."""
    wn = 0
    n = len(ring)
    for i in range(n):
        ax, ay = ring[i]
        bx, by = ring[(i + 1) % n]
        if ay <= qy:
            if by > qy and (bx - ax) * (qy - ay) - (qx - ax) * (by - ay) > 0:
                wn += 1
        else:
            if by <= qy and (bx - ax) * (qy - ay) - (qx - ax) * (by - ay) < 0:
                wn -= 1
    return wn


def positive_union_outer(ring_in: Sequence[Pt]) -> List[Pt]:
    """Clipper `Execute(ctUnion, pftPositive, pftPositive)` on ONE closed path,
    restated geometrically: the boundary of {winding > 0}, the loop that holds
    the lowest (max-y) vertex (Clipper's first OutRec), intersection vertices
    rounded half-away like Clipper's IntersectPoint, duplicate and collinear
    vertices dropped (FixupOutPolygon), orientation positive, and rotated so it
    ENDS at the top-most vertex (the last local maximum is added to the front
    of the OutPt ring and BuildResult walks Prev from Pts->Prev)."""
    ring: List[Pt] = []
    for p in ring_in:
        if not ring or ring[-1] != p:
            ring.append(p)
    while len(ring) > 1 and ring[0] == ring[-1]:
        ring.pop()
    n = len(ring)
    if n < 3:
        return []
    # split points per segment: (t, key) ; key identifies the node
    splits: List[List[Tuple[Fraction, Tuple[Fraction, Fraction]]]] = []
    for i in range(n):
        a, b = ring[i], ring[(i + 1) % n]
        splits.append([(Fraction(0), (Fraction(a[0]), Fraction(a[1]))),
                       (Fraction(1), (Fraction(b[0]), Fraction(b[1])))])
    for i in range(n):
        a, b = ring[i], ring[(i + 1) % n]
        for j in range(i + 1, n):
            c, d = ring[j], ring[(j + 1) % n]
            r = _seg_intersection(a, b, c, d)
            if r is None:
                continue
            t, u, pt = r
            if 0 < t < 1:
                splits[i].append((t, pt))
            if 0 < u < 1:
                splits[j].append((u, pt))
    eps = Fraction(1, 1 << 60)
    edges = []  # (start_node, end_node)
    for i in range(n):
        a, b = ring[i], ring[(i + 1) % n]
        sp = sorted(set(splits[i]))
        dx, dy = b[0] - a[0], b[1] - a[1]
        for k in range(len(sp) - 1):
            (t0, n0), (t1, n1) = sp[k], sp[k + 1]
            if n0 == n1:
                continue
            mx = (n0[0] + n1[0]) / 2
            my = (n0[1] + n1[1]) / 2
            # right side of the directed edge = (dy, -dx)
            w_right = _winding(ring, mx + eps * dy, my - eps * dx)
            w_left = _winding(ring, mx - eps * dy, my + eps * dx)
            if (w_left > 0) != (w_right > 0):
                if w_left > 0:
                    edges.append((n0, n1))
                else:
                    edges.append((n1, n0))
    if not edges:
        return []
    out_map = {}
    for e in edges:
        out_map.setdefault(e[0], []).append(e)
    used = set()
    loops = []
    for e in edges:
        if e in used:
            continue
        loop = []
        cur = e
        while cur not in used:
            used.add(cur)
            loop.append(cur[0])
            cands = [c for c in out_map.get(cur[1], []) if c not in used]
            if not cands:
                break
            if len(cands) == 1:
                cur = cands[0]
            else:
                # touching node: take the most clockwise (right-most) turn so
                # the outer boundary stays one loop
                inx = float(cur[1][0] - cur[0][0])
                iny = float(cur[1][1] - cur[0][1])

                def turn(c):
                    ox = float(c[1][0] - c[0][0])
                    oy = float(c[1][1] - c[0][1])
                    return math.atan2(inx * oy - iny * ox, inx * ox + iny * oy)
                cur = min(cands, key=turn)
        loops.append(loop)
    # Clipper's first OutRec starts at the lowest local minimum (largest Y)
    loops.sort(key=lambda lp: (-max(p[1] for p in lp), min(p[0] for p in lp)))
    loop = loops[0]

    def rnd(fr: Fraction) -> int:
        return _cround(float(fr))
    pts = [(rnd(p[0]), rnd(p[1])) for p in loop]
    # drop duplicates and collinear vertices (FixupOutPolygon)
    changed = True
    while changed and len(pts) >= 3:
        changed = False
        m = len(pts)
        for i in range(m):
            p, c, nx = pts[i - 1], pts[i], pts[(i + 1) % m]
            if c == nx or c == p or (c[1] - p[1]) * (nx[0] - c[0]) == (c[0] - p[0]) * (nx[1] - c[1]):
                del pts[i]
                changed = True
                break
    if len(pts) < 3:
        return []
    if _shoelace2(pts) < 0:
        pts.reverse()
    top = min(range(len(pts)), key=lambda i: (pts[i][1], -pts[i][0]))
    return pts[top + 1:] + pts[:top + 1]


def expand_polygon(points: Sequence[Pt], factor: float = 2.0):
    """polygon.rs:51-56 -> clip_polygon(.., Expand): Option<Vec<Point<i32>>>."""
    d = offset_distance(points, factor)
    raw = raw_offset_ring(points, d)
    res = positive_union_outer(raw)
    return res if res else None


# --------------------------------------------------------------------------
# get_polygons_from_bitmap / get_boxes_and_box_scores  (metrics.rs:37-127)
# --------------------------------------------------------------------------
def _round_half_away(v: float) -> float:
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def _as_u32(v: float) -> int:
    if v != v or v <= 0:
        return 0
    return int(min(v, 4294967295.0))


class DegeneratePolygon(RuntimeError):
    """expand_polygon returned None; the reference unwraps it (metrics.rs:103) and aborts."""


def get_polygons_from_bitmap(pred: np.ndarray, bitmap: np.ndarray, adj: Sequence[float],
                             box_thresh: float = 0.7, min_size: float = 5.0,
                             unclip: float = 2.0, skip_degenerate: bool = False):
    """pred: HxW float, bitmap: HxW u8 (0/1).  Returns (polygons, scores)."""
    image = (bitmap.astype(np.int64) * 255).astype(np.uint8)  # metrics.rs:69
    contours = find_contours(image)
    boxes, scores = [], []
    for contour in contours:
        epsilon = 0.01 * arc_length(contour, True)
        if epsilon == 0.0:
            epsilon = 0.01
        points = approximate_polygon_dp(contour, epsilon, True)
        if len(points) > 1 and points[0] == points[-1]:
            points.pop()
        if len(points) < 4:
            continue
        score = box_score_fast(pred, points)
        if box_thresh > score:
            continue
        expanded = expand_polygon(points, unclip)
        if expanded is None:
            if skip_degenerate:
                continue
            raise DegeneratePolygon("expand_polygon returned None (reference unwraps -> abort)")
        _, sside = get_min_area_bounding_box(expanded)
        if sside < min_size:
            continue
        boxes.append([(_as_u32(_round_half_away(p[0] / adj[0])),
                       _as_u32(_round_half_away(p[1] / adj[1]))) for p in expanded])
        scores.append(score)
    return boxes, scores


def get_boxes_and_box_scores(pred: np.ndarray, adjust_values: np.ndarray, thresh: float = 0.6,
                             skip_degenerate: bool = False):
    """pred: Nx1xHxW float32, adjust_values: Nx2 float64 -> (polygons[N], scores[N])."""
    seg = binarize(pred, thresh)
    polys, scores = [], []
    for b in range(pred.shape[0]):
        p, s = get_polygons_from_bitmap(pred[b, 0], seg[b, 0], adjust_values[b], skip_degenerate=skip_degenerate)
        polys.append(p)
        scores.append(s)
    return polys, scores

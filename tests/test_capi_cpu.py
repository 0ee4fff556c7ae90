"""CPU-side checks of the product library: it loads, exports every symbol that
include/ocr_amd.h declares, refuses to compute without a GPU, and its host C++
geometry reproduces the reference KATs (no oracle involved in the product path)."""
import os
import random
import re

import numpy as np
import pytest
from PIL import Image

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import postproc_oracle as O
from tests import kat_postproc as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "ocr_amd.h")).read()
    declared = set(re.findall(r"\b(ocr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.ocr_version().decode().startswith("ocr_amd")
    assert L.ocr_rec_alphabet().decode() == "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"


def test_product_library_ships_no_test_hooks():
    """libocr_amd.so exports the C ABI of include/ocr_amd.h and NOTHING else: no ocr_test_* hook, no C++ internal
    (exports.map).  The hooks live in libocr_amd_test.so, a library of its own built from the same objects."""
    import subprocess
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TWVBDR"}
    assert exported(capi.LIB_PATH) == set(capi.EXPORTS)                # exactly the header's surface, nothing else
    hooks = {n for n in exported(capi.TEST_LIB_PATH) if n.startswith("ocr_test_")}
    assert hooks


def test_no_cpu_fallback_without_gpu():
    if capi.lib().ocr_device_count() > 0:
        pytest.skip("a GPU is visible")
    blob = W.pack_blob(W.make_rec_weights(0))
    with pytest.raises(capi.OcrError) as e:
        capi.Recognizer(blob, 0)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_bad_blob_rejected_before_gpu():
    h = capi.C.c_void_p()
    code = capi.lib().ocr_det_create(b"nope", 4, 0, capi.C.byref(h))
    assert code != 0 and not h


def test_default_params_are_reference_constants():
    p = capi.PostprocParams()
    capi.lib().ocr_postproc_default_params(capi.C.byref(p))
    assert (p.thresh, p.box_thresh, p.min_size, p.unclip_ratio) == (0.6, 0.7, 5.0, 2.0)   # metrics.rs:38,64,66,103
    assert p.skip_degenerate == 0


def test_host_min_area_box_kat():
    box, sside = capi.host_min_area_box(K.MIN_AREA_BOX_IN)
    assert box == K.MIN_AREA_BOX_OUT
    assert abs(sside - K.MIN_AREA_BOX_SSIDE) < np.finfo(np.float64).eps


def _bitmap(golden_dir, name):
    return (np.array(Image.open(os.path.join(golden_dir, name)).convert("L")) // 255).astype(np.uint8)


def test_host_geometry_reproduces_img55_kat(golden_dir):
    bm = _bitmap(golden_dir, "gt_shrinked_img55.png")
    cands = capi.host_contour_candidates(bm)
    assert len(cands) == 4
    out = []
    for c in cands:
        ex, sside = capi.host_expand_polygon(c, 2.0)
        assert sside >= 5.0
        out.append(ex)
    assert out == K.IMG55_POLYS_ADJ1      # metrics.rs:519-568 (adj = 1)


@pytest.mark.parametrize("name", ["gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"])
def test_host_geometry_matches_oracle_on_other_fixtures(golden_dir, name):
    bm = _bitmap(golden_dir, name)
    cands = capi.host_contour_candidates(bm)
    ocands = []
    for c in O.find_contours(bm * 255):
        eps = 0.01 * O.arc_length(c, True) or 0.01
        pts = O.approximate_polygon_dp(c, eps, True)
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts.pop()
        if len(pts) >= 4:
            ocands.append(pts)
    assert cands == ocands
    for c in cands:
        assert capi.host_expand_polygon(c, 2.0)[0] == O.expand_polygon(c, 2.0)


def test_host_geometry_matches_oracle_on_random_blobs():
    rng = np.random.RandomState(5)
    # noise blobs: many tiny contours, holes, border-touching shapes
    f = rng.rand(96, 128)
    for _ in range(3):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5
    bm = (f > np.median(f)).astype(np.uint8)
    cands = capi.host_contour_candidates(bm)
    ocands = []
    for c in O.find_contours(bm * 255):
        eps = 0.01 * O.arc_length(c, True) or 0.01
        pts = O.approximate_polygon_dp(c, eps, True)
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts.pop()
        if len(pts) >= 4:
            ocands.append(pts)
    assert len(cands) > 3 and cands == ocands
    for c in cands:
        assert capi.host_expand_polygon(c, 2.0)[0] == (O.expand_polygon(c, 2.0) or [])


def test_expand_random_star_polygons_match_oracle():
    rnd = random.Random(11)
    import math
    for _ in range(60):
        k = rnd.randint(4, 14)
        cx, cy = rnd.randint(200, 400), rnd.randint(200, 400)
        pts = []
        for i in range(k):
            ang = 2 * math.pi * i / k
            r = rnd.uniform(15, 120)
            pts.append((int(cx + r * math.cos(ang)), int(cy + r * math.sin(ang))))
        if len(set(pts)) < len(pts):
            continue
        assert capi.host_expand_polygon(pts, 2.0)[0] == (O.expand_polygon(pts, 2.0) or [])


def test_expand_word_boxes_simple_ring_fast_path_matches_oracle():
    """Word-sized boxes - what the unclip step sees almost always: their offset ring is simple and takes the union's fast path
    (postproc_geom.cpp::positive_union_outer).  Both orientations, slanted quads, a straight (collinear) vertex on an edge,
    pentagons and thin slivers against the oracle's general arrangement."""
    rnd = random.Random(23)
    cases = []
    for _ in range(150):
        x0, y0 = rnd.randint(20, 500), rnd.randint(20, 500)
        w, h = rnd.randint(3, 120), rnd.randint(2, 40)
        sl = rnd.randint(-8, 8)
        quad = [(x0, y0), (x0 + w, y0 + sl), (x0 + w, y0 + h + sl), (x0, y0 + h)]
        kind = rnd.randint(0, 3)
        if kind == 1:
            quad = quad[::-1]                                           # the other orientation
        elif kind == 2:
            quad.insert(1, (x0 + w // 2, y0 + (sl * (w // 2)) // max(w, 1)))   # a (nearly) straight vertex on the top edge
        elif kind == 3:
            quad.insert(2, (x0 + w + rnd.randint(1, 9), y0 + sl + h // 2))     # a pentagon
        if len(set(quad)) == len(quad):
            cases.append(quad)
    assert len(cases) > 100
    for c in cases:
        assert capi.host_expand_polygon(c, 2.0)[0] == (O.expand_polygon(c, 2.0) or []), c


def test_cpp_host_mirror_compiles_and_reports_errors(tmp_path):
    """ocr-rs_amd/host/ocr_rs.hpp (the C++ mirror of the reference's call sites) builds with
    plain g++ against the C ABI and surfaces failures as ocr_rs::Error, not as a crash."""
    import subprocess
    host = os.path.join(ROOT, "ocr-rs_amd", "host")
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "demo")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(host, "demo.cpp"), "-L" + libdir,
                           "-locr_amd", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, "/nonexistent.ocrw", "/nonexistent.ocrw"], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "doesn't exist" in r.stderr      # text_detection/mod.rs:36-39 wording
    det = tmp_path / "det.ocrw"
    rec = tmp_path / "rec.ocrw"
    det.write_bytes(W.pack_blob(W.make_det_weights(0)))
    rec.write_bytes(W.pack_blob(W.make_rec_weights(0)))
    r = subprocess.run([exe, str(det), str(rec)], env=env, capture_output=True, text=True)
    if capi.lib().ocr_device_count() == 0:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
    else:
        assert r.returncode == 0 and "classified as" in r.stdout


@pytest.mark.parametrize("h,w,density,seed", [(37, 45, 0.55, 1), (64, 64, 0.5, 2), (9, 131, 0.6, 3), (50, 33, 0.35, 4),
                                              (1, 70, 0.5, 5), (70, 1, 0.5, 6), (40, 96, 0.97, 7)])
def test_run_based_border_following_matches_the_oracle_on_noise(h, w, density, seed):
    """The product's tracer walks horizontal runs of a packed bit image instead of testing every pixel; on noise
    (holes inside blobs, blobs touching every edge, widths that are no multiple of 32) it must report exactly the
    borders the per-pixel Suzuki-Abe restatement of the oracle reports, in the same order."""
    rng = np.random.RandomState(seed)
    bm = (rng.rand(h, w) < density).astype(np.uint8)
    got = capi.host_contour_candidates(bm)
    want = []   # the oracle's pieces: contours -> DP (epsilon 1 % of the closed arc length) -> >= 4 points
    for c in O.find_contours(bm):
        eps = 0.01 * O.arc_length(c, True)
        pts = O.approximate_polygon_dp(c, eps if eps != 0.0 else 0.01, True)
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts = pts[:-1]
        if len(pts) >= 4:
            want.append([(int(p[0]), int(p[1])) for p in pts])
    assert got == want


def test_device_hypot_restatement_is_libm_hypot():
    """unclip.hip sums a polygon's perimeter with glibc's hypot restated for the device (csrc/hypot_glibc.hpp): geo's euclidean_length calls
    libm's hypot (/root/reference/src/polygon.rs:27), which differs from sqrt(dx^2 + dy^2) by an ulp on 0.6 % of integer pairs.  Host code
    only: every integer pair up to 3 000 through both, no mismatch - and sqrt(dx^2 + dy^2) would not pass this."""
    import ctypes as C
    L = capi.test_lib()
    L.ocr_test_hypot_port_mismatches.restype = C.c_longlong
    assert L.ocr_test_hypot_port_mismatches(3000) == 0
    import math
    libm = C.CDLL("libm.so.6")
    libm.hypot.restype = C.c_double
    libm.hypot.argtypes = [C.c_double, C.c_double]
    assert any(libm.hypot(float(a), float(b)) != math.sqrt(a * a + b * b) for a in range(1, 300) for b in range(1, a))

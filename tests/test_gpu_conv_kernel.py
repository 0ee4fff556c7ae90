"""Kernel-level parity of conv_igemm (the dominant kernel) through the ocr_test_conv_run hook:
one launch on caller data against ATen-CPU conv2d on the same operands.

f32: relative 2e-5 of the layer's scale (accumulation order + FMA contraction only).
bf16 (OCR_PRECISION_BF16): operands are rounded to bf16 on the way in, so the reference is computed from
the same rounded operands with f32 accumulation; what is left is accumulation order, which can move a
result across a bf16 rounding boundary: at most one bf16 ulp, at a small fraction of the elements.
(End-to-end bit-level emulation of the bf16 network is not a usable bar: the stack of roundings is chaotic -
a 1e-6 relative change of the input frame moves the emulated map by 9e-3, see tests/test_gpu_bf16.py.)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    d = capi.Detector(W.pack_blob(W.make_det_weights(0)), 0)
    yield d
    d.close()


def _q(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


def _ref(x, wg, stride, scale, bias, residual, up_residual, relu):
    """x: N H W C, wg: O (kh kw) I -> (out, out2) NHWC, f32 accumulate."""
    cout, kk, cin = wg.shape
    ks = int(round(kk ** 0.5))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.from_numpy(wg).reshape(cout, ks, ks, cin).permute(0, 3, 1, 2)
    y = F.conv2d(xt, wt, None, stride, (ks - 1) // 2)
    if scale is not None:
        y = y * torch.from_numpy(scale).view(1, -1, 1, 1)
    if bias is not None:
        y = y + torch.from_numpy(bias).view(1, -1, 1, 1)
    if residual is not None:
        y = y + torch.from_numpy(residual).permute(0, 3, 1, 2)
    if relu:
        y = F.relu(y)
    y2 = None
    if up_residual is not None:
        u = F.interpolate(torch.from_numpy(up_residual).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
        y2 = (y + u).permute(0, 2, 3, 1).contiguous().numpy()
    return y.permute(0, 2, 3, 1).contiguous().numpy(), y2


def _check(got, ref, bf16_out):
    scale = float(np.abs(ref).max()) + 1e-12
    if not bf16_out:
        assert float(np.abs(got - ref).max()) / scale < 2e-5
        return
    refq = _q(ref)
    # one bf16 ulp of the reference value (8 significand bits) + f32 accumulation slack near zero
    tol = np.abs(refq) * 2.0 ** -7 + 1e-5 * scale
    diff = np.abs(got - refq)
    assert (diff <= tol).all(), float((diff - tol).max())
    assert float((diff > 1e-5 * scale).mean()) < 0.01   # flips are rare


CASES = [
    # n, h, w, cin, cout, ks, stride, bn, residual, up, relu, want_out
    (1, 10, 14, 64, 64, 3, 1, True, True, False, True, True),       # basic_block conv2 + identity, ragged M
    (2, 18, 22, 64, 128, 3, 2, True, False, False, True, True),     # strided conv1 of layer2.0, odd output grid
    (2, 18, 22, 64, 128, 1, 2, True, False, False, False, True),    # downsample 1x1 s2
    (1, 12, 20, 128, 256, 1, 1, False, False, True, False, True),   # lateral in3 + top-down sum (two outputs)
    (1, 12, 20, 64, 256, 1, 1, False, False, True, False, False),   # lateral in2: only the sum is stored
    (1, 9, 7, 256, 64, 3, 1, False, False, False, False, True),     # out2..5
    (1, 12, 12, 512, 256, 1, 1, False, False, False, False, True),  # in5
    (4, 256, 256, 64, 64, 3, 1, True, True, False, True, True),     # full-size tile (128x64) and XCD remap
    (8, 128, 128, 128, 256, 1, 1, False, False, True, False, True), # 128x128 tile with both outputs
    # 3x3 stride-1 convs with 128 / 256 output channels: in bf16 the tap-sharing loop (one staged run of pixels for the three taps
    # of a kernel row, out-of-row pixels zeroed in registers) on 128 x 128 and 128 x 64 tiles
    (2, 20, 20, 256, 256, 3, 1, True, True, False, True, True),     # layer4-like: tiles span image rows and the image boundary
    (3, 5, 3, 128, 128, 3, 1, True, False, False, False, True),     # rows of 3 pixels: a run wraps a row every third pixel
    (1, 1, 1, 64, 128, 3, 1, False, False, False, False, True),     # a single pixel: every tap but the centre is padding
    (2, 33, 17, 128, 256, 3, 1, True, True, False, True, True),     # odd grid, M = 1122: partial last tile
    (1, 40, 128, 128, 128, 3, 1, False, False, False, True, True),  # rows as long as a tile: every tile starts at a row start
    (1, 7, 130, 64, 192, 3, 1, True, False, False, True, True),     # rows of BM + 2 pixels, Cout = 192 (128 x 64 tiles)
]


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_conv_matches_aten(det, case, bf16):
    n, h, w, cin, cout, ks, stride, bn, has_res, has_up, relu, want_out = case
    rng = np.random.default_rng(hash(case[:7]) & 0xFFFF)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wg = (rng.standard_normal((cout, ks * ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    pad = (ks - 1) // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    up = rng.standard_normal((n, ho // 2, wo // 2, cout), dtype=np.float32) if has_up else None
    if bf16:
        x, wg = _q(x), _q(wg)
        res = _q(res) if res is not None else None
        up = _q(up) if up is not None else None
    out, out2 = det.debug_conv_run(x, wg, stride, scale, bias, res, up, relu, None, bf16, bf16, want_out, has_up)
    ref, ref2 = _ref(x, wg, stride, scale, bias, res, up, relu)
    if want_out:
        _check(out, ref, bf16)
    if has_up:
        _check(out2, ref2, bf16)


X3_CASES = [
    # n, h, w, cin, cout, ks, stride, bn, residual, relu
    (2, 18, 22, 64, 128, 3, 2, True, False, True),      # strided conv1 of layer2.0, odd output grid, ragged M (128 x 128 tile)
    (1, 10, 14, 64, 64, 3, 1, True, True, True),        # 128 x 64 tile, residual, M < one tile
    (3, 40, 40, 256, 512, 3, 2, True, False, True),     # layer4.0 conv1: K = 2304
    (1, 9, 7, 256, 64, 3, 1, False, False, False),      # out2..5 shape, no epilogue terms
    (36, 10, 10, 512, 512, 1, 1, False, False, False),  # the Winograd GEMMs of layer4 (K = 512), M = 3600: partial last tile
    (4, 128, 128, 128, 256, 3, 2, True, False, True),   # many tiles: XCD remap, both tile shapes by Cout
]


@pytest.mark.parametrize("case", X3_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_conv_split_bf16_matches_aten(det, case):
    """The f32 conv on the bf16 matrix cores (operands split into three bf16 terms, six partial products, f32
    accumulate) is held to the SAME bar as the exact-f32 MFMA kernel: 2e-5 of the layer's scale
    against ATen's f32 conv on the same f32 operands."""
    n, h, w, cin, cout, ks, stride, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:7]) & 0xFFFF)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wg = (rng.standard_normal((cout, ks * ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    pad = (ks - 1) // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    out, _ = det.debug_conv_run(x, wg, stride, scale, bias, res, None, relu, None, False, False, True, False, variant=2)
    ref, _ = _ref(x, wg, stride, scale, bias, res, None, relu)
    _check(out, ref, False)
    # and against the exact-f32 MFMA kernel on the same operands: the two agree far inside the bar
    out32, _ = det.debug_conv_run(x, wg, stride, scale, bias, res, None, relu, None, False, False, True, False)
    assert float(np.abs(out - out32).max()) / (float(np.abs(ref).max()) + 1e-12) < 5e-6


WIDE_CASES = [
    # n, h, w, cin, cout, ks, stride, bn, residual, relu      (the launches conv_x3w.hip takes: NHWC store, Cout % 128 == 0, >= 256 strips x column tiles)
    (4, 160, 160, 64, 128, 3, 2, True, False, True),     # layer2.0 conv1: one column tile, 400 strips, K = 576
    (9, 90, 90, 128, 256, 3, 2, True, True, True),       # two column tiles, odd grid, ragged last strip (M = 18 225), residual
    (3, 80, 80, 256, 512, 3, 2, True, False, True),      # layer4.0 conv1 shape: four column tiles, K = 2 304, 75 strips: one-strip tiles, idle workgroups
    (4, 160, 160, 64, 128, 1, 2, True, False, False),    # the downsample of layer2.0: two K-steps per tile
    (2, 100, 100, 256, 256, 1, 1, False, False, False),  # in4-like lateral: 1x1 s1
    (5, 64, 64, 32, 128, 1, 1, True, True, True),        # ONE K-step per tile: the flat sequence crosses a tile boundary at every step
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "x".join(str(v) for v in c[:7]))
def test_wide_split_bf16_form_is_bit_identical(det, case):
    """conv_x3w.hip (256 x 128 tiles, one persistent workgroup per CU, K-steps as one flat sequence across its tiles) computes the same
    products in the same order as conv_igemm's 128-wide split-bf16 tiles: the SAME BITS, and the usual 2e-5 against ATen."""
    n, h, w, cin, cout, ks, stride, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:7]) & 0xFFFF)
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wg = (rng.standard_normal((cout, ks * ks, cin), dtype=np.float32) / np.sqrt(ks * ks * cin)).astype(np.float32)
    pad = (ks - 1) // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    wide, _ = det.debug_conv_run(x, wg, stride, scale, bias, res, None, relu, None, False, False, True, False, variant=3)
    out, _ = det.debug_conv_run(x, wg, stride, scale, bias, res, None, relu, None, False, False, True, False, variant=2)
    assert np.array_equal(wide, out)
    ref, _ = _ref(x, wg, stride, scale, bias, res, None, relu)
    _check(wide, ref, False)


@pytest.mark.parametrize("shape", [(36, 3200, 256, 256), (36, 800, 512, 512), (16, 1000, 128, 384)], ids=lambda s: "x".join(map(str, s)))
def test_wide_split_bf16_batched_gemms_are_bit_identical(det, shape):
    """the Winograd GEMMs of layer3 / layer4 (36 problems of [tiles x Cin] . [Cin x Cout]) and a ragged one: a tile never spans two
    problems, the last strip of a problem is partly empty"""
    b, m, k, nn = shape
    rng = np.random.default_rng(b * 1000 + m)
    x = rng.standard_normal((b, m, k), dtype=np.float32)
    wg = (rng.standard_normal((b, nn, k), dtype=np.float32) / np.sqrt(k)).astype(np.float32)
    wide = det.debug_gemm_batched(x, wg, variant=3)
    out = det.debug_gemm_batched(x, wg, variant=2)
    assert np.array_equal(wide, out)
    ref = np.einsum("bmk,bnk->bmn", x.astype(np.float64), wg.astype(np.float64))
    assert float(np.abs(wide - ref).max()) / float(np.abs(ref).max()) < 2e-5


def test_conv_split_bf16_wide_exponents(det):
    """Operands spread over 24 binades and values that need all 24 significand bits: the three-term split is exact
    (no term under- or overflows in bf16's f32-sized exponent range), so the bar does not move."""
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 1, 16, 16, 64, 64
    x = (rng.standard_normal((n, h, w, cin)) * np.exp2(rng.integers(-12, 12, (n, h, w, cin)))).astype(np.float32)
    wg = (rng.standard_normal((cout, 9, cin)) * np.exp2(rng.integers(-12, 12, (cout, 9, cin))) / 24.0).astype(np.float32)
    out, _ = det.debug_conv_run(x, wg, 1, None, None, None, None, False, None, False, False, True, False, variant=2)
    xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
    wt = torch.from_numpy(wg).double().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xt, wt, None, 1, 1).permute(0, 2, 3, 1).numpy()
    mag = F.conv2d(xt.abs(), wt.abs(), None, 1, 1).permute(0, 2, 3, 1).numpy()   # sum |a b|: the scale f32 rounding works on
    assert float((np.abs(out - ref) / mag).max()) < 4e-6


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_cat4_conv_matches_aten(det, bf16):
    """bin_conv1 over the virtual concat [up8(p5), up4(p4), up2(p3), p2] (model.rs:139-146): the kernel
    gathers the four levels itself; bf16 operands still give an f32 result (it feeds the f32 head)."""
    n, h, w = 2, 16, 24
    rng = np.random.default_rng(5)
    lv = [rng.standard_normal((n, h >> s, w >> s, 64), dtype=np.float32) for s in (3, 2, 1, 0)]   # p5, p4, p3, p2
    wg = (rng.standard_normal((64, 9, 256), dtype=np.float32) / 48.0).astype(np.float32)
    scale = 0.5 + rng.random(64, dtype=np.float32)
    bias = rng.standard_normal(64, dtype=np.float32)
    if bf16:
        lv, wg = [_q(a) for a in lv], _q(wg)
    flat = np.concatenate([a.ravel() for a in lv])
    out, _ = det.debug_conv_run(flat, wg, 1, scale, bias, None, None, True, (n, h, w), bf16, False)
    ups = [np.repeat(np.repeat(a, 8 >> i, axis=1), 8 >> i, axis=2) for i, a in enumerate(lv)]
    fuse = np.concatenate(ups, axis=3)
    ref, _ = _ref(fuse, wg, 1, scale, bias, None, None, True)
    _check(out, ref, False)


WINO_CASES = [
    # n, h, w, cin, cout, bn, residual, relu
    (2, 20, 20, 512, 512, True, True, True),     # layer4 conv2 at 640x640
    (1, 40, 40, 256, 256, True, False, True),    # layer3 conv1
    (1, 25, 25, 256, 256, True, True, True),     # odd grid (800x800 frames: layer4 is 25x25)
    (3, 3, 5, 256, 512, False, False, False),    # smaller than a tile row, ragged both ways, Cin != Cout
    (1, 1, 1, 256, 256, True, True, False),      # a single pixel: every tap but the centre is padding
    # few channels
    (2, 16, 32, 64, 64, True, True, True),       # whole blocks
    (1, 9, 21, 64, 64, True, False, True),       # ragged blocks, odd sizes
    (3, 1, 1, 64, 64, False, False, False),      # single pixels
    (2, 160, 160, 64, 64, True, True, True),     # layer1 grid at 640x640
    (2, 80, 80, 128, 128, True, True, True),     # layer2 conv2: four channel chunks, two output-channel blocks
    (1, 13, 37, 128, 64, False, True, False),    # p3 lateral term (128 -> 64), ragged blocks
    (1, 8, 16, 64, 128, True, False, True),      # 64 -> 128
]



@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd_conv_matches_aten(det, case):
    """Winograd F(2x2,3x3) path (weight transform, input transform, batched GEMM, output transform with the
    epilogue) against ATen's direct conv2d on the same operands: relative 2e-5 of the layer's scale."""
    n, h, w, cin, cout, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:5]) & 0xFFFF)
    x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)   # post-ReLU-like activations
    wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, h, w, cout), dtype=np.float32) if has_res else None
    got = det.debug_winograd_conv(x, wg, scale, bias, res, relu)
    ref, _ = _ref(x, wg, 1, scale, bias, res, None, relu)
    _check(got, ref, False)


W43_CASES = [
    (2, 20, 20, 512, 512, True, True, True),     # layer4 grid at 640x640: 5 x 5 tiles per image
    (1, 40, 40, 256, 256, True, True, True),     # layer3 grid
    (1, 9, 21, 256, 256, True, False, True),     # ragged tiles: 9 = 2 * 4 + 1, 21 = 5 * 4 + 1
    (3, 3, 5, 256, 512, False, False, False),    # smaller than a tile one way, ragged the other, Cin != Cout
    (1, 1, 1, 512, 512, True, True, False),      # a single pixel
    (2, 6, 7, 64, 64, False, True, True),        # few channels
]


@pytest.mark.parametrize("case", W43_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd43_conv_matches_aten(det, case):
    """Winograd F(4x4,3x3) path (36-component weight transform, input transform, batched GEMM, output transform with the
    epilogue) against ATen's direct conv2d on the same operands.  Its transforms multiply by up to 8 and cancel more than
    F(2x2)'s, so the bar is 1e-4 of the layer's scale (measured: about 1e-5)."""
    n, h, w, cin, cout, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:5]) & 0xFFFF)
    x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
    wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, h, w, cout), dtype=np.float32) if has_res else None
    got = det.debug_winograd_conv(x, wg, scale, bias, res, relu, unfused=3)
    ref, _ = _ref(x, wg, 1, scale, bias, res, None, relu)
    err = float(np.abs(got - ref).max()) / (float(np.abs(ref).max()) + 1e-12)
    assert err < 1e-4, err


W43F_CASES = [
    (2, 16, 32, 64, 64, True, True, True),       # whole 16 x 16 blocks
    (1, 9, 21, 64, 64, True, False, True),       # ragged blocks and tiles
    (3, 1, 1, 64, 64, False, False, False),      # single pixels
    (2, 160, 160, 64, 64, True, True, True),     # layer1 grid at 640x640
    (2, 80, 80, 128, 128, True, True, True),     # layer2: eight channel chunks, two output-channel blocks
    (1, 13, 37, 128, 64, False, True, False),    # p3 lateral term (128 -> 64), ragged
    (1, 8, 16, 64, 128, True, False, True),      # 64 -> 128
    (5, 24, 40, 64, 64, True, True, False),
    (1, 40, 40, 256, 64, False, False, False),   # out4 (256 -> 64): sixteen channel chunks
    (2, 24, 48, 256, 128, True, True, True),     # 256 input channels, two output-channel blocks share a patch
    (40, 16, 16, 64, 64, True, True, True),      # 40 blocks: the eight XCD runs of 5, several rounds on a small grid
    (1, 64, 272, 64, 64, True, True, False),     # 68 blocks in one image: runs that split image rows
]


@pytest.mark.parametrize("case", W43F_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd43_fused_conv_matches_aten(det, case):
    """The fused F(4x4,3x3) kernel (winograd43_fused.hip) against ATen's direct conv2d, same bar as the unfused F(4x4) path."""
    n, h, w, cin, cout, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:5]) & 0xFFFF)
    x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
    wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, h, w, cout), dtype=np.float32) if has_res else None
    got = det.debug_winograd_conv(x, wg, scale, bias, res, relu, unfused=4)
    ref, _ = _ref(x, wg, 1, scale, bias, res, None, relu)
    err = float(np.abs(got - ref).max()) / (float(np.abs(ref).max()) + 1e-12)
    assert err < 1e-4, err


W43X_CASES = W43F_CASES + [
    (3, 16, 16, 64, 64, True, True, True),       # an odd number of pixel blocks: the last workgroup owns one block and an empty one
    (1, 16, 16, 128, 128, False, True, False),   # a single block, two output-channel blocks
    (2, 33, 17, 64, 64, True, True, True),       # one pixel past a block edge both ways
]


@pytest.mark.parametrize("case", W43X_CASES, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd43_fused_x3_conv_matches_aten(det, case):
    """The fused F(4x4,3x3) kernel with its GEMMs on the bf16 matrix cores (winograd43_x3.hip: f32 operands as three bf16 terms,
    six partial products, f32 accumulate) against ATen's direct conv2d: the SAME bar as the f32 kernels."""
    n, h, w, cin, cout, bn, has_res, relu = case
    rng = np.random.default_rng(hash(case[:5]) & 0xFFFF)
    x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
    wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    scale = (0.5 + rng.random(cout, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(cout, dtype=np.float32) if bn else None
    res = rng.standard_normal((n, h, w, cout), dtype=np.float32) if has_res else None
    got = det.debug_winograd_conv(x, wg, scale, bias, res, relu, unfused=5)
    ref, _ = _ref(x, wg, 1, scale, bias, res, None, relu)
    err = float(np.abs(got - ref).max()) / (float(np.abs(ref).max()) + 1e-12)
    assert err < 1e-4, err


def test_winograd43_fused_x3_matches_f32_kernel(det):
    """Split-bf16 against the exact-f32 fused kernel on the same operands, with 24 binades of exponent spread across pixels and
    channels: the split is exact and the dropped products are below 2^-23, so the two differ by summation order only."""
    rng = np.random.default_rng(7)
    n, h, w, cin, cout = 2, 32, 48, 64, 64
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32) * np.exp2(rng.integers(-12, 12, (n, h, w, 1))).astype(np.float32)
    wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
    a = det.debug_winograd_conv(x, wg, None, None, None, False, unfused=4)
    b = det.debug_winograd_conv(x, wg, None, None, None, False, unfused=5)
    # compare per 16 x 16 block (a tile's error scales with the largest value the tile's transform saw)
    for yy in range(0, h, 16):
        for xx in range(0, w, 16):
            sa, sb = a[:, yy:yy + 16, xx:xx + 16], b[:, yy:yy + 16, xx:xx + 16]
            assert float(np.abs(sa - sb).max()) <= 2e-5 * (float(np.abs(x[:, max(yy - 1, 0):yy + 17, max(xx - 1, 0):xx + 17]).max()) + 1e-30), (yy, xx)


@pytest.mark.parametrize("num_cus", [1, 5, 37, 100, 304])
def test_winograd43_fused_x3_any_grid_size(det, num_cus):
    """winograd43_x3.hip's persistent grid is num_cus workgroups: whatever the grid, every pair of blocks is computed exactly once."""
    rng = np.random.default_rng(num_cus)
    for n, h, w, cin, cout in ((3, 40, 56, 64, 64), (2, 24, 40, 128, 128)):
        x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
        wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
        res = rng.standard_normal((n, h, w, cout), dtype=np.float32)
        got = det.debug_winograd_conv(x, wg, None, None, res, True, unfused=5 | (num_cus << 8))
        ref, _ = _ref(x, wg, 1, None, None, res, None, True)
        assert float(np.abs(got - ref).max()) / (float(np.abs(ref).max()) + 1e-12) < 1e-4


@pytest.mark.parametrize("num_cus", [1, 5, 37, 100, 304])
def test_winograd43_fused_any_grid_size(det, num_cus):
    """The fused kernel's persistent grid is 2 x num_cus workgroups (multiProcessorCount of the device: a partitioned or
    differently sized part gives another number): whatever the grid - fewer workgroups than XCDs, not a multiple of
    eight, more than there are blocks - every block is computed exactly once."""
    rng = np.random.default_rng(num_cus)
    for n, h, w, cin, cout in ((3, 40, 56, 64, 64), (2, 24, 40, 128, 128)):
        x = np.maximum(rng.standard_normal((n, h, w, cin), dtype=np.float32), 0)
        wg = (rng.standard_normal((cout, 9, cin), dtype=np.float32) / np.sqrt(9 * cin)).astype(np.float32)
        res = rng.standard_normal((n, h, w, cout), dtype=np.float32)
        got = det.debug_winograd_conv(x, wg, None, None, res, True, unfused=4 | (num_cus << 8))
        ref, _ = _ref(x, wg, 1, None, None, res, None, True)
        assert float(np.abs(got - ref).max()) / (float(np.abs(ref).max()) + 1e-12) < 1e-4


@pytest.mark.parametrize("shape", [(2, 16, 32), (1, 9, 21), (3, 1, 1), (2, 160, 160), (1, 8, 16), (5, 24, 40), (1, 64, 272)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("has_res,relu,bn", [(True, True, True), (False, False, False), (True, False, True)])
def test_bf16_c64_patch_conv_matches_aten(det, shape, has_res, relu, bn):
    """conv3x3_bf16_c64.hip (bf16 3x3 s1 64 -> 64: patch staged once, weights in registers) against ATen on the same
    bf16-rounded operands: at most one bf16 ulp, like every other bf16 kernel."""
    n, h, w = shape
    rng = np.random.default_rng(hash(shape) & 0xFFFF)
    x = _q(np.maximum(rng.standard_normal((n, h, w, 64), dtype=np.float32), 0))
    wg = _q((rng.standard_normal((64, 9, 64), dtype=np.float32) / np.sqrt(9 * 64)).astype(np.float32))
    scale = (0.5 + rng.random(64, dtype=np.float32)) if bn else None
    bias = rng.standard_normal(64, dtype=np.float32) if bn else None
    res = _q(rng.standard_normal((n, h, w, 64), dtype=np.float32)) if has_res else None
    out, _ = det.debug_conv_run(x, wg, 1, scale, bias, res, None, relu, None, True, True, True, False, variant=1)
    ref, _ = _ref(x, wg, 1, scale, bias, res, None, relu)
    _check(out, ref, True)



BLOCK_SHAPES = [(2, 16, 32), (1, 9, 21), (3, 1, 1), (2, 160, 160), (1, 8, 16), (5, 24, 40), (1, 64, 272), (1, 7, 15), (2, 17, 33), (1, 200, 200)]


def _block_operands(seed, n, h, w, bn=True):
    rng = np.random.default_rng(seed)
    x = _q(np.maximum(rng.standard_normal((n, h, w, 64), dtype=np.float32), 0))
    w1 = _q((rng.standard_normal((64, 9, 64), dtype=np.float32) / np.sqrt(9 * 64)).astype(np.float32))
    w2 = _q((rng.standard_normal((64, 9, 64), dtype=np.float32) / np.sqrt(9 * 64)).astype(np.float32))
    s1 = (0.5 + rng.random(64, dtype=np.float32)) if bn else None
    b1 = rng.standard_normal(64, dtype=np.float32) if bn else None
    s2 = (0.5 + rng.random(64, dtype=np.float32)) if bn else None
    b2 = rng.standard_normal(64, dtype=np.float32) if bn else None
    return x, w1, w2, s1, b1, s2, b2


@pytest.mark.parametrize("shape", BLOCK_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("bn", [True, False])
def test_bf16_basic_block_one_launch_equals_two_launches_bit_for_bit(det, shape, bn):
    """basic_block_bf16_c64.hip (model.rs:40-55 as ONE launch, the activation between the convs in LDS) against the same block as two
    conv3x3_bf16_c64 launches: the same bits - at the image edge (the intermediate's zero padding), ragged blocks, one-pixel images,
    the full 160 x 160 of configs[1]."""
    n, h, w = shape
    x, w1, w2, s1, b1, s2, b2 = _block_operands(hash(shape) & 0xFFFF, n, h, w, bn)
    two, _ = det.debug_bf16_basic_block(x, w1, w2, s1, b1, s2, b2, fused=False)
    one, _ = det.debug_bf16_basic_block(x, w1, w2, s1, b1, s2, b2, fused=True)
    assert np.isfinite(one).all() and float(np.abs(two).max()) > 0.1
    assert np.array_equal(one, two)


@pytest.mark.parametrize("num_cus", [1, 3, 8, 24, 100, 256, 512, 1000])
def test_bf16_basic_block_any_persistent_grid(det, num_cus):
    """the persistent grid is one workgroup per CU; whatever its size (fewer than the XCDs, not a multiple of eight, oversubscribed for
    head_cus_yield, more workgroups than blocks) every block is computed exactly once"""
    x, w1, w2, s1, b1, s2, b2 = _block_operands(num_cus, 3, 40, 56)
    two, _ = det.debug_bf16_basic_block(x, w1, w2, s1, b1, s2, b2, fused=False)
    one, _ = det.debug_bf16_basic_block(x, w1, w2, s1, b1, s2, b2, fused=True, num_cus=num_cus)
    assert np.array_equal(one, two)


def test_bf16_basic_block_matches_aten(det):
    """... and against ATen on the same bf16-rounded operands, the intermediate rounded to bf16 as the kernel rounds it: one bf16 ulp"""
    x, w1, w2, s1, b1, s2, b2 = _block_operands(5, 2, 24, 40)
    got, _ = det.debug_bf16_basic_block(x, w1, w2, s1, b1, s2, b2, fused=True)
    mid, _ = _ref(x, w1, 1, s1, b1, None, None, True)
    ref, _ = _ref(_q(mid), w2, 1, s2, b2, x, None, True)
    refq = _q(ref)
    scale = float(np.abs(ref).max())
    # one bf16 ulp of the result + what a one-ulp flip of an intermediate value (ATen's accumulation order is not the kernel's) moves it by
    assert (np.abs(got - refq) <= np.abs(refq) * 2.0 ** -7 + 1e-3 * scale).all()
    assert float((got != refq).mean()) < 0.02

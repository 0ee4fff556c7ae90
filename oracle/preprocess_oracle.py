"""ORACLE (test infrastructure, never shipped): CPU restatement of

    preprocess_image(file, (W, H)) -> (GrayImage W x H, adjust_x, adjust_y)
                                   /root/reference/src/image_ops.rs:188-220

whose arithmetic lives in the un-vendored crate image 0.23.11 (Cargo.lock:607-608):
`DynamicImage::resize(w, h, FilterType::Triangle)` (aspect preserving: resize_dimensions, then
imageops::resize = vertical_sample followed by horizontal_sample, f32 weights, each pass truncated to
u8), `.to_luma()` (0.2126 R + 0.7152 G + 0.0722 B in f32, truncated), then zero padding to W x H.

Pinning: the reference's KAT (image_ops.rs:805-1008) starts from JPEG files decoded by jpeg-decoder
0.1.20; this container only has libjpeg (PIL), whose IDCT/upsampling differs in the last bit, so the
fixtures test_data/preprocessed_img*.png can only be matched approximately from here
(tests/test_oracle_preprocess.py: >= 94 % of pixels identical, max |diff| <= 2; adjust values
exact).  PARITY PARTIALLY PINNED.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def resize_dimensions(width: int, height: int, nwidth: int, nheight: int):
    """image 0.23 `resize_dimensions(.., fill=false)`: fit inside nwidth x nheight, keep aspect."""
    ratio = width * nheight
    nratio = nwidth * height
    use_width = nratio <= ratio
    inter = (height * nwidth) // width if use_width else (width * nheight) // height
    inter = max(1, inter)
    return (nwidth, inter) if use_width else (inter, nheight)


def _weights(in_size: int, out_size: int):
    """Per output index: (left, normalised f32 triangle weights) exactly as sample.rs computes them."""
    ratio = F(in_size) / F(out_size)
    sratio = ratio if ratio >= F(1.0) else F(1.0)
    src_support = F(1.0) * sratio            # Triangle support = 1.0
    res = []
    for o in range(out_size):
        inp = F(F(o) + F(0.5)) * ratio
        left = int(math.floor(float(F(inp - src_support))))
        left = min(max(left, 0), in_size - 1)
        right = int(math.ceil(float(F(inp + src_support))))
        right = min(max(right, left + 1), in_size)
        inp = F(inp - F(0.5))
        ws = []
        s = F(0.0)
        for i in range(left, right):
            x = F(F(F(i) - inp) / sratio)
            ax = abs(x)
            w = F(F(1.0) - ax) if ax < F(1.0) else F(0.0)
            ws.append(w)
            s = F(s + w)
        ws = [F(w / s) for w in ws]
        res.append((left, ws))
    return res


def _to_u8(t: np.ndarray) -> np.ndarray:
    # image 0.23.11: NumCast::from(clamp(t, 0, max)) - truncation.  (The rounding variant was tried
    # against the reference fixtures and matches only ~30 % of the pixels; truncation matches 95-99 %.)
    return np.clip(t, F(0.0), F(255.0)).astype(np.uint8)


def _sample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """vertical_sample for an H x W x C u8 image (axis 0); horizontal = same on the transposed image."""
    tab = _weights(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.float32)
    for o, (left, ws) in enumerate(tab):
        t = np.zeros(img.shape[1:], np.float32)
        for k, w in enumerate(ws):                      # t += pixel * w, sequential f32
            t = (t + src[left + k] * w).astype(np.float32)
        out[o] = _to_u8(t)
    return out


def resize_triangle(rgba: np.ndarray, nwidth: int, nheight: int) -> np.ndarray:
    tmp = _sample_axis0(rgba, nheight)                                   # vertical_sample
    return _sample_axis0(tmp.transpose(1, 0, 2), nwidth).transpose(1, 0, 2)   # horizontal_sample


def to_luma(rgba: np.ndarray) -> np.ndarray:
    r, g, b = (rgba[..., k].astype(np.float32) for k in range(3))
    l = (F(0.2126) * r + F(0.7152) * g).astype(np.float32)
    l = (l + F(0.0722) * b).astype(np.float32)
    return l.astype(np.uint8)                            # NumCast: truncation


def preprocess_image(rgba: np.ndarray, target_w: int, target_h: int):
    """rgba: H x W x 4 u8 (decoded image).  Returns (gray target_h x target_w u8, adjust_x, adjust_y)."""
    h, w = rgba.shape[:2]
    nw, nh = resize_dimensions(w, h, target_w, target_h)
    gray = to_luma(np.ascontiguousarray(resize_triangle(rgba, nw, nh)))
    out = np.zeros((target_h, target_w), np.uint8)
    out[:nh, :nw] = gray
    return out, nw / w, nh / h

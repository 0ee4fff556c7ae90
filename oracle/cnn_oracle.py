"""ORACLE (test infrastructure, never shipped): the two reference CNN graphs composed
from the plain-C operators of oracle/cnn_oracle.c.

    det_forward  <- /root/reference/src/text_detection/model.rs:65-152 (resnet18, eval)
    rec_forward  <- /root/reference/src/char_recognition/model.rs:27-39
    rec_classify <- /root/reference/src/char_recognition/mod.rs:53-56, utils.rs:28-43

PARITY UNPINNED by the reference's own tests (no weights / activations exist
there); pinned here to ATen through oracle/torch_ref.py (tests/test_oracle_cnn.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_F = ctypes.POINTER(ctypes.c_float)


def build() -> str:
    so = os.path.join(_HERE, "libcnn_oracle.so")
    src = os.path.join(_HERE, "cnn_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcnn_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(_F)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x, w, stride, pad, bias=None):
    x, w = _f32(x), _f32(w)
    n, c, h, wd = x.shape
    co, _, k, _ = w.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = np.empty((n, co, ho, wo), np.float32)
    b = _f32(bias) if bias is not None else None
    _lib().orc_conv2d(_p(x), n, c, h, wd, _p(w), co, k, stride, pad, _p(b) if b is not None else None, _p(y))
    return y


def batch_norm(x, p, prefix):
    n, c, h, w = x.shape
    _lib().orc_batch_norm(_p(x), n, c, h * w, _p(_f32(p[prefix + ".weight"])), _p(_f32(p[prefix + ".bias"])),
                          _p(_f32(p[prefix + ".running_mean"])), _p(_f32(p[prefix + ".running_var"])),
                          ctypes.c_float(1e-5))
    return x


def relu(x):
    _lib().orc_relu(_p(x), ctypes.c_size_t(x.size))
    return x


def add(x, y):
    y = _f32(y)
    _lib().orc_add(_p(x), _p(y), ctypes.c_size_t(x.size))
    return x


def max_pool2d(x, k, s, pad):
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    y = np.empty((n, c, ho, wo), np.float32)
    _lib().orc_max_pool2d(_p(x), n, c, h, w, k, s, pad, _p(y))
    return y


def upsample(x, k):
    n, c, h, w = x.shape
    y = np.empty((n, c, h * k, w * k), np.float32)
    _lib().orc_upsample_nearest(_p(x), n * c, h, w, k, _p(y))
    return y


def conv_transpose2d(x, w, bias):
    x, w, bias = _f32(x), _f32(w), _f32(bias)
    n, c, h, wd = x.shape
    co = w.shape[1]
    y = np.empty((n, co, 2 * h, 2 * wd), np.float32)
    _lib().orc_conv_transpose2d_k2s2(_p(x), n, c, h, wd, _p(w), co, _p(bias), _p(y))
    return y


def sigmoid(x):
    _lib().orc_sigmoid(_p(x), ctypes.c_size_t(x.size))
    return x


def _basic_block(x, p, prefix, stride):  # model.rs:40-55
    y = relu(batch_norm(conv2d(x, p[prefix + ".conv1.weight"], stride, 1), p, prefix + ".bn1"))
    y = batch_norm(conv2d(y, p[prefix + ".conv2.weight"], 1, 1), p, prefix + ".bn2")
    if (prefix + ".downsample.0.weight") in p:  # model.rs:30-38
        d = batch_norm(conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0), p, prefix + ".downsample.1")
    else:
        d = x
    return relu(add(y, d))


def det_forward(p: Dict[str, np.ndarray], x: np.ndarray, stages: dict | None = None) -> np.ndarray:
    """model.rs:107-151, train=false."""
    x = _f32(x)
    s = relu(batch_norm(conv2d(x, p["conv1.weight"], 2, 3), p, "bn1"))   # :108-111
    s = max_pool2d(s, 3, 2, 1)                                            # :112
    if stages is not None:
        stages["stem"] = s.copy()
    feats = []
    cur = s
    for li in range(1, 5):                                                # :113-120
        cur = _basic_block(cur, p, f"layer{li}.0", 1 if li == 1 else 2)
        cur = _basic_block(cur, p, f"layer{li}.1", 1)
        feats.append(cur)
        if stages is not None:
            stages[f"layer{li}"] = cur.copy()
    x1, x2, x3, x4 = feats
    i2 = conv2d(x1, p["in2.weight"], 1, 0)                                # :115
    i3 = conv2d(x2, p["in3.weight"], 1, 0)                                # :118
    i4 = conv2d(x3, p["in4.weight"], 1, 0)                                # :121
    i5 = conv2d(x4, p["in5.weight"], 1, 0)                                # :123
    p2 = conv2d(add(upsample(i3, 2), i2), p["out2.weight"], 1, 1)         # :126-129
    p3 = upsample(conv2d(add(upsample(i4, 2), i3), p["out3.weight"], 1, 1), 2)   # :130-133
    p4 = upsample(conv2d(add(upsample(i5, 2), i4), p["out4.weight"], 1, 1), 4)   # :134-137
    p5 = upsample(conv2d(i5, p["out5.weight"], 1, 1), 8)                  # :138
    fuse = np.concatenate([p5, p4, p3, p2], axis=1)                       # :140
    if stages is not None:
        stages["fuse"] = fuse.copy()
    y = relu(batch_norm(conv2d(fuse, p["bin_conv1.weight"], 1, 1), p, "bin_bn1"))  # :143-145
    if stages is not None:
        stages["bin1"] = y.copy()
    y = relu(batch_norm(conv_transpose2d(y, p["bin_conv_tr1.weight"], p["bin_conv_tr1.bias"]), p, "bin_bn2"))
    y = conv_transpose2d(y, p["bin_conv_tr2.weight"], p["bin_conv_tr2.bias"])       # :149
    if stages is not None:
        stages["logit"] = y.copy()
    return sigmoid(y)                                                     # :150


def rec_forward(p: Dict[str, np.ndarray], x: np.ndarray) -> np.ndarray:
    """char_recognition/model.rs:27-39, train=false (dropout is the identity)."""
    t = _f32(x).reshape(-1, 1, 28, 28)
    t = max_pool2d(conv2d(t, p["conv1.weight"], 1, 0, p["conv1.bias"]), 2, 2, 0)
    t = max_pool2d(conv2d(t, p["conv2.weight"], 1, 0, p["conv2.bias"]), 2, 2, 0)
    t = np.ascontiguousarray(t.reshape(-1, 1024))
    n = t.shape[0]
    h = np.empty((n, 512), np.float32)
    _lib().orc_linear(_p(t), n, 1024, _p(_f32(p["fc1.weight"])), 512, _p(_f32(p["fc1.bias"])), _p(h))
    relu(h)
    o = np.empty((n, 62), np.float32)
    _lib().orc_linear(_p(h), n, 512, _p(_f32(p["fc2.weight"])), 62, _p(_f32(p["fc2.bias"])), _p(o))
    return o


def rec_classify(logits: np.ndarray):
    logits = _f32(logits)
    n = logits.shape[0]
    lab = np.empty(n, np.int32)
    prob = np.empty(n, np.float64)
    _lib().orc_softmax_top1(_p(logits), n, logits.shape[1], lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                            prob.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return lab, prob

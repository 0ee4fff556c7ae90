"""ORACLE (test infrastructure, never shipped, never the thing measured except as
bench.py's cpu_baseline).

Both CNN graphs of the reference restated operator-for-operator on PyTorch-CPU.
The reference executes these graphs through tch 0.3.0 -> libtorch 1.7.0 ATen CPU
kernels (Cargo.toml:18, Dockerfile:7); `torch.nn.functional` on CPU calls the same
ATen operators (conv2d, batch_norm, max_pool2d, upsample_nearest2d,
conv_transpose2d, sigmoid, linear, softmax, topk), so this is the closest
executable stand-in for the reference in an image without Rust.

PARITY UNPINNED by the reference itself: it ships no weights and no test holds an
expected activation.  This restatement is pinned only to
oracle/cnn_oracle.c (independent plain-C loops) and to the committed goldens.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

VALUES = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"  # utils.rs:7


def _t(params: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}


def _bn(x, p, prefix):
    # tch nn::batch_norm2d defaults: eps 1e-5, eval mode uses running stats
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], False, 0.1, 1e-5)


def _basic_block(x, p, prefix, stride):
    """model.rs:40-55"""
    y = F.conv2d(x, p[prefix + ".conv1.weight"], None, stride, 1)
    y = F.relu(_bn(y, p, prefix + ".bn1"))
    y = F.conv2d(y, p[prefix + ".conv2.weight"], None, 1, 1)
    y = _bn(y, p, prefix + ".bn2")
    if (prefix + ".downsample.0.weight") in p:  # model.rs:30-38
        d = F.conv2d(x, p[prefix + ".downsample.0.weight"], None, stride, 0)
        d = _bn(d, p, prefix + ".downsample.1")
    else:
        d = x
    return F.relu(y + d)


def det_forward(params: Dict[str, np.ndarray], x: np.ndarray, stages: dict | None = None) -> np.ndarray:
    """resnet18(..).forward_t(xs, train=false), model.rs:107-151.
    x: N x 1 x H x W f32 (raw 0..255) -> N x 1 x H x W probabilities."""
    p = _t(params)
    with torch.no_grad():
        xs = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        s = F.conv2d(xs, p["conv1.weight"], None, 2, 3)
        s = F.relu(_bn(s, p, "bn1"))
        s = F.max_pool2d(s, 3, 2, 1, 1, False)
        if stages is not None:
            stages["stem"] = s.numpy().copy()
        feats = []
        cur = s
        for li in range(1, 5):
            cur = _basic_block(cur, p, f"layer{li}.0", 1 if li == 1 else 2)
            cur = _basic_block(cur, p, f"layer{li}.1", 1)
            feats.append(cur)
            if stages is not None:
                stages[f"layer{li}"] = cur.numpy().copy()
        x1, x2, x3, x4 = feats
        i2 = F.conv2d(x1, p["in2.weight"])
        i3 = F.conv2d(x2, p["in3.weight"])
        i4 = F.conv2d(x3, p["in4.weight"])
        i5 = F.conv2d(x4, p["in5.weight"])

        def up(t, k):
            return F.interpolate(t, scale_factor=k, mode="nearest")

        p2 = F.conv2d(up(i3, 2) + i2, p["out2.weight"], None, 1, 1)
        p3 = up(F.conv2d(up(i4, 2) + i3, p["out3.weight"], None, 1, 1), 2)
        p4 = up(F.conv2d(up(i5, 2) + i4, p["out4.weight"], None, 1, 1), 4)
        p5 = up(F.conv2d(i5, p["out5.weight"], None, 1, 1), 8)
        fuse = torch.cat([p5, p4, p3, p2], 1)
        if stages is not None:
            stages["fuse"] = fuse.numpy().copy()
        y = F.relu(_bn(F.conv2d(fuse, p["bin_conv1.weight"], None, 1, 1), p, "bin_bn1"))
        if stages is not None:
            stages["bin1"] = y.numpy().copy()
        y = F.conv_transpose2d(y, p["bin_conv_tr1.weight"], p["bin_conv_tr1.bias"], 2, 0)
        y = F.relu(_bn(y, p, "bin_bn2"))
        y = F.conv_transpose2d(y, p["bin_conv_tr2.weight"], p["bin_conv_tr2.bias"], 2, 0)
        if stages is not None:
            stages["logit"] = y.numpy().copy()
        return torch.sigmoid(y).numpy()


def det_forward_bf16(params: Dict[str, np.ndarray], x: np.ndarray) -> np.ndarray:
    """The OCR_PRECISION_BF16 arithmetic of the product, restated (there is no reference behaviour to
    match: the reference is f32 only; BASELINE config 5 names bf16 as an optional precision):
    every conv from conv1 to bin_conv_tr1 takes operands rounded to bf16 (activations and weights),
    accumulates in f32, applies folded batch norm / residual / ReLU in f32 and stores bf16; the FPN
    top-down sums are stored bf16 too; bin_conv_tr1's result, bin_bn2, bin_conv_tr2 and the sigmoid are f32.
    Differences to the product: accumulation order only (then a bf16 rounding may flip by one ulp)."""
    p = _t(params)

    def q(t):
        return t.to(torch.bfloat16).to(torch.float32)

    def conv(t, name, stride=1, pad=0):
        return F.conv2d(t, q(p[name]), None, stride, pad)

    def block(t, prefix, stride):
        y = q(F.relu(_bn(conv(t, prefix + ".conv1.weight", stride, 1), p, prefix + ".bn1")))
        y = _bn(conv(y, prefix + ".conv2.weight", 1, 1), p, prefix + ".bn2")
        if (prefix + ".downsample.0.weight") in p:
            d = q(_bn(conv(t, prefix + ".downsample.0.weight", stride, 0), p, prefix + ".downsample.1"))
        else:
            d = t
        return q(F.relu(y + d))

    with torch.no_grad():
        xs = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        s = F.conv2d(q(xs), q(p["conv1.weight"]), None, 2, 3)
        s = q(F.max_pool2d(F.relu(_bn(s, p, "bn1")), 3, 2, 1, 1, False))
        feats = []
        cur = s
        for li in range(1, 5):
            cur = block(cur, f"layer{li}.0", 1 if li == 1 else 2)
            cur = block(cur, f"layer{li}.1", 1)
            feats.append(cur)
        x1, x2, x3, x4 = feats

        def up(t, k):
            return F.interpolate(t, scale_factor=k, mode="nearest")

        # laterals: the raw in3..in5 are stored bf16; each sum is formed from the f32 conv result plus
        # the bf16-stored upper lateral and stored bf16 (conv_igemm's out2)
        i5 = q(conv(x4, "in5.weight"))
        i4f = conv(x3, "in4.weight")
        i3f = conv(x2, "in3.weight")
        i2f = conv(x1, "in2.weight")
        i4, i3 = q(i4f), q(i3f)
        s4 = q(i4f + up(i5, 2))
        s3 = q(i3f + up(i4, 2))
        s2 = q(i2f + up(i3, 2))
        p2 = q(conv(s2, "out2.weight", 1, 1))
        p3 = up(q(conv(s3, "out3.weight", 1, 1)), 2)
        p4 = up(q(conv(s4, "out4.weight", 1, 1)), 4)
        p5 = up(q(conv(i5, "out5.weight", 1, 1)), 8)
        fuse = torch.cat([p5, p4, p3, p2], 1)
        y = q(F.relu(_bn(conv(fuse, "bin_conv1.weight", 1, 1), p, "bin_bn1")))
        y = F.conv_transpose2d(y, q(p["bin_conv_tr1.weight"]), p["bin_conv_tr1.bias"], 2, 0)
        y = F.relu(_bn(y, p, "bin_bn2"))
        y = F.conv_transpose2d(y, p["bin_conv_tr2.weight"], p["bin_conv_tr2.bias"], 2, 0)
        return torch.sigmoid(y).numpy()


def rec_forward(params: Dict[str, np.ndarray], x: np.ndarray) -> np.ndarray:
    """Net::forward_t(xs, train=false), char_recognition/model.rs:27-39 -> N x 62 logits."""
    p = _t(params)
    with torch.no_grad():
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).view(-1, 1, 28, 28)
        t = F.conv2d(t, p["conv1.weight"], p["conv1.bias"])
        t = F.max_pool2d(t, 2)
        t = F.conv2d(t, p["conv2.weight"], p["conv2.bias"])
        t = F.max_pool2d(t, 2)
        t = t.reshape(-1, 1024)
        t = F.relu(F.linear(t, p["fc1.weight"], p["fc1.bias"]))
        t = F.linear(t, p["fc2.weight"], p["fc2.bias"])
        return t.numpy()


def rec_classify(logits: np.ndarray):
    """char_recognition/mod.rs:53-56 + utils.rs:28-43: softmax(-1, Double), top-1."""
    t = torch.from_numpy(logits).softmax(-1, dtype=torch.float64)
    v, i = t.topk(1, -1, True, True)
    return i[:, 0].numpy().astype(np.int32), v[:, 0].numpy()

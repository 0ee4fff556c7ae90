"""Host-side mirror of the reference's text-detection call sites, over the C ABI.

    reference (Rust)                                     here
    ---------------------------------------------------  ---------------------------------
    resnet18(&vs.root()); vs.load(file)   mod.rs:35-44   net = resnet18(weights_blob, device)
    net.forward_t(&x, false)              mod.rs:52-54   net.forward_t(x, train=False)
    get_boxes_and_box_scores(&pred, &adj) metrics.rs:37  get_boxes_and_box_scores(net, pred, adj)
    PolygonScores{polygons, scores}       metrics.rs:32  PolygonScores(polygons, scores)

Tensors are numpy arrays (host) or torch CUDA tensors (device); results are plain
Python lists of (x, y) u32 vertices in original-image coordinates and f64 scores.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import capi

DEFAULT_WIDTH = 800    # text_detection/mod.rs:20
DEFAULT_HEIGHT = 800   # text_detection/mod.rs:21


@dataclass
class PolygonScores:            # metrics.rs:32-35
    polygons: List[List[List[Tuple[int, int]]]]
    scores: List[List[float]]


class FuncT:
    """What `resnet18(&nn::Path) -> FuncT<'static>` returns (model.rs:154-156)."""

    def __init__(self, weights_blob: bytes, device: int = 0, options: str = ""):
        self._det = capi.Detector(weights_blob, device, options=options) if options else capi.Detector(weights_blob, device)

    @property
    def handle(self) -> capi.Detector:
        return self._det

    def forward_t(self, xs, train: bool = False):
        """`ModuleT::forward_t(&self, xs, train)`: xs N x 1 x H x W f32 -> N x 1 x H x W."""
        if train:
            raise capi.OcrError(1, "inference-only build: forward_t(train=true) is the reference's training path")
        if isinstance(xs, np.ndarray):
            if xs.ndim != 4 or xs.shape[1] != 1:
                raise capi.OcrError(1, f"expected N x 1 x H x W, got {xs.shape}")
            if xs.dtype == np.uint8:     # the reference's GrayImage bytes (image_ops.rs:350-364): converted inside the first kernel
                return self._det.forward_host_u8(xs)
            return self._det.forward_host(xs)
        import torch
        if not (xs.is_cuda and xs.dtype == torch.float32 and xs.is_contiguous() and xs.dim() == 4 and xs.shape[1] == 1):
            raise capi.OcrError(1, "expected a contiguous CUDA f32 tensor N x 1 x H x W")
        n, _, h, w = xs.shape
        out = torch.empty_like(xs)
        cur = torch.cuda.current_stream(xs.device)
        if cur.cuda_stream:
            # a real stream: the forward is enqueued on it, ordered after whatever produced xs and before
            # whatever the caller enqueues next on the same stream (torch semantics)
            self._det.set_stream(cur.cuda_stream)
            self._det.forward_device(xs.data_ptr(), n, h, w, out.data_ptr())
        else:
            # torch's default stream is the legacy null stream (handle 0), which the C ABI reads as "the handle's
            # own stream" - a hipStreamNonBlocking one that the null stream does not order against.  Make the
            # call blocking on both sides instead: pending producers of xs first, the forward before returning.
            cur.synchronize()
            self._det.set_stream(None)
            self._det.forward_device(xs.data_ptr(), n, h, w, out.data_ptr())
            self._det.synchronize()
        return out

    def close(self):
        self._det.close()


def resnet18(weights_blob: bytes, device: int = 0, options: str = "") -> FuncT:
    """`options`: engine options of include/ocr_amd.h (the reference has none; "" = the defaults)."""
    return FuncT(weights_blob, device, options)


def preprocess_image(net: FuncT, rgba: np.ndarray, target_dim=(DEFAULT_WIDTH, DEFAULT_HEIGHT)):
    """image_ops::preprocess_image (image_ops.rs:188-220) on an already decoded RGBA image:
    returns (GrayImage as H x W u8, adjust_x, adjust_y)."""
    return net.handle.preprocess_image(rgba, target_dim[0], target_dim[1])


def get_boxes_and_box_scores(net: FuncT, pred, adjust_values, skip_degenerate: bool = False) -> PolygonScores:
    """metrics.rs:37-56.  `net` supplies the GPU/stream the HIP post-processing kernels run on.
    Raises OcrError(code 6) where the reference would abort on `expand_polygon(..).unwrap()`."""
    adj = np.ascontiguousarray(adjust_values, dtype=np.float64)
    if isinstance(pred, np.ndarray):
        p = np.ascontiguousarray(pred, dtype=np.float32)
        n, _, h, w = p.shape
        polys, scores = net.handle.postprocess(p, n, h, w, adj, capi.MEM_HOST,
                                               capi.default_params(skip_degenerate))
    else:
        import torch
        n, _, h, w = pred.shape
        torch.cuda.current_stream().synchronize()
        polys, scores = net.handle.postprocess(pred, n, h, w, adj, capi.MEM_DEVICE,
                                               capi.default_params(skip_degenerate))
    return PolygonScores(polys, scores)

"""Host-side mirror of the reference's character-recognition call sites, over the C ABI.

    Net::new(&weights.root()); weights.load(file)   char_recognition/mod.rs:44-46
    net.forward_t(&image_tensor, false)             mod.rs:53-54, model.rs:27-39
    .softmax(-1, Kind::Double); topk(&res, 1)       mod.rs:55-56, utils.rs:28-43
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import capi

VALUES = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"   # utils.rs:7
VALUES_COUNT = len(VALUES)


class Net:
    def __init__(self, weights_blob: bytes, device: int = 0):
        self._rec = capi.Recognizer(weights_blob, device)

    @property
    def handle(self) -> capi.Recognizer:
        return self._rec

    def forward_t(self, xs: np.ndarray, train: bool = False) -> np.ndarray:
        """xs.view([-1,1,28,28]) ... fc2: N x 784 (or N x 1 x 28 x 28) -> N x 62 logits."""
        if train:
            raise capi.OcrError(1, "inference-only build: dropout/backward are the reference's training path")
        xs = np.ascontiguousarray(xs, dtype=np.float32)
        if xs.size % 784 != 0:
            raise capi.OcrError(1, f"shape {xs.shape} is invalid for view([-1, 1, 28, 28])")   # tch would error too
        return self._rec.forward_host(xs.reshape(-1, 784))

    def predict(self, xs: np.ndarray) -> List[Tuple[str, float]]:
        """run_prediction's tail: softmax(-1, f64) + topk(.., 1)[0] per crop -> (char, probability)."""
        labels, probs = self._rec.classify_host(np.ascontiguousarray(xs, dtype=np.float32).reshape(-1, 784))
        return [(VALUES[int(i)], float(p)) for i, p in zip(labels, probs)]

    def ctc_greedy_decode(self, logits: np.ndarray, blank: int = VALUES_COUNT) -> List[str]:
        """EXTENSION (the reference has no sequence recogniser): CTC best-path decode of N x T x C logits over the reference's alphabet
        (utils.rs:7: classes 0..61 = VALUES) plus a blank (default: class 62) -> one string per crop."""
        labels, lengths = self._rec.ctc_greedy_decode(np.ascontiguousarray(logits, dtype=np.float32), blank)
        out = []
        for row, n in zip(labels, lengths):
            out.append("".join(VALUES[int(k)] if 0 <= int(k) < VALUES_COUNT else "?" for k in row[:int(n)]))
        return out

    def close(self):
        self._rec.close()


def topk1(probabilities: np.ndarray) -> Tuple[str, float]:
    """utils::topk(tensor, 1)[0] for one softmax row of 62 probabilities."""
    p = np.asarray(probabilities).reshape(-1)
    if p.shape[0] != VALUES_COUNT:
        raise ValueError(f"unexpected tensor shape {p.shape}")                       # utils.rs:33 panics
    i = int(np.argmax(p))
    return VALUES[i], float(p[i])

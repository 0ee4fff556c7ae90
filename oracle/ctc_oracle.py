"""ORACLE (test infrastructure, never shipped, never measured as the product).

CTC greedy (best-path) decoding, exact integer arithmetic - the checker of ocr_ctc_greedy_decode (ocr-rs_amd/csrc/ctc.hip).  An EXTENSION
with no counterpart in the reference: BASELINE.json's north_star / configs[2] name a "32x128 recognition + CTC greedy decode" stage, while
/root/reference/src/char_recognition/model.rs:27-39 classifies single 28 x 28 glyphs (alphabet: /root/reference/src/utils.rs:7-9, 62
characters; a CTC head over it has 62 + 1 blank = 63 classes).  The rule is the published one (Graves et al. 2006, best-path decoding):

    a[t]   = the first class attaining the maximum of column t            (ties: lowest class index)
    labels = a with consecutive repeats collapsed, then blanks removed    (i.e. keep a[t] iff a[t] != blank and (t == 0 or a[t] != a[t-1]))

Parity unpinned by the reference (it holds no such test); pinned by the hand-written vectors of tests/test_ctc.py.
"""
from __future__ import annotations

import numpy as np


def ctc_greedy_decode(logits: np.ndarray, blank: int):
    """logits N x T x C -> (labels N x T int32 padded with -1, lengths N int32)."""
    x = np.asarray(logits)
    n, t, c = x.shape
    assert 0 <= blank < c
    a = np.argmax(x, axis=2)          # numpy: the first maximum
    labels = np.full((n, t), -1, np.int32)
    lengths = np.zeros(n, np.int32)
    for i in range(n):
        k, prev = 0, -1
        for j in range(t):
            v = int(a[i, j])
            if v != blank and v != prev:
                labels[i, k] = v
                k += 1
            prev = v
        lengths[i] = k
    return labels, lengths

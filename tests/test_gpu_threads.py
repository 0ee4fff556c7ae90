"""Handles are per thread, the library is shared: two host threads, each with its own detector and recogniser handle on the same
GPU (own streams, own post-processing pools), run the whole pipeline at the same time - a serving process with several request
threads, or one thread per GPU (include/ocr_amd.h: a handle is not thread-safe, different handles are independent).  Every
result must be bit-identical to the same work done alone: no shared scratch, no library-wide state, thread-local errors."""
import threading

import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from tests.test_gpu_config4 import run_pipeline

pytestmark = pytest.mark.gpu


def _same(a, b):
    pa, la, sa, ca, ba = a
    pb, lb, sb, cb, bb = b
    return (np.array_equal(pa, pb) and la == lb and len(sa) == len(sb)
            and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(sa, sb)) and np.array_equal(ca, cb) and np.array_equal(ba, bb))


def test_two_threads_two_handle_pairs_same_gpu_bit_identical():
    det_w, rec_w = W.make_det_weights_text(), W.make_rec_weights(0)
    blob_d, blob_r = W.pack_blob(det_w), W.pack_blob(rec_w)
    pages = [W.synth_text_pages(100 + t, 6, 640, 640)[0] for t in range(2)]
    precisions = [capi.PRECISION_F32, capi.PRECISION_BF16]   # one thread in each precision: different kernels side by side
    # alone, one after the other
    alone = []
    for t in range(2):
        det, rec = capi.Detector(blob_d, 0, options="post_threads=2"), capi.Recognizer(blob_r, 0)
        alone.append(run_pipeline(det, rec, pages[t], precisions[t]))
        det.close()
        rec.close()
    # together
    results, errors = [None, None], []
    start = threading.Barrier(2)

    def work(t):
        try:
            det, rec = capi.Detector(blob_d, 0, options="post_threads=2"), capi.Recognizer(blob_r, 0)
            start.wait()
            for _ in range(4):
                got = run_pipeline(det, rec, pages[t], precisions[t])
                if results[t] is not None and not _same(results[t], got):
                    errors.append(f"thread {t}: results changed between rounds")
                results[t] = got
            # an error on this thread stays on this thread
            with pytest.raises(capi.OcrError):
                det.forward_host(np.zeros((1, 1, 33, 32), np.float32))
            det.close()
            rec.close()
        except Exception as e:   # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        assert _same(alone[t], results[t]), f"thread {t}: concurrent result differs from the same work done alone"

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


def test_destroy_with_async_work_on_the_callers_stream():
    """`ocr_det_set_stream` + `ocr_det_forward_async` / `ocr_rec_classify_async`, then destroy WITHOUT synchronising: the handle
    drains the caller's stream before it frees the workspace those launches use - the results are complete and correct."""
    import torch
    det_w, rec_w = W.make_det_weights(0), W.make_rec_weights(0)
    x = torch.from_numpy(W.synth_image_batch(3, 8, 320, 320)).cuda()
    crops = torch.from_numpy(W.synth_crops(4, 2000)).cuda()
    ref_det, ref_rec = capi.Detector(W.pack_blob(det_w), 0), capi.Recognizer(W.pack_blob(rec_w), 0)
    want = torch.empty_like(x)
    ref_det.forward_device(x.data_ptr(), 8, 320, 320, want.data_ptr())
    ref_det.synchronize()
    want_lab = torch.empty(2000, dtype=torch.int32, device="cuda")
    want_pr = torch.empty(2000, dtype=torch.float64, device="cuda")
    ref_rec.classify_device(crops.data_ptr(), 2000, 0, want_lab.data_ptr(), want_pr.data_ptr())
    ref_rec.synchronize()
    ref_det.close()
    ref_rec.close()
    stream = torch.cuda.Stream()
    for _ in range(3):
        det, rec = capi.Detector(W.pack_blob(det_w), 0), capi.Recognizer(W.pack_blob(rec_w), 0)
        det.set_stream(stream.cuda_stream)
        rec.set_stream(stream.cuda_stream)
        got = torch.zeros_like(x)
        lab = torch.zeros(2000, dtype=torch.int32, device="cuda")
        pr = torch.zeros(2000, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):   # a queue of work behind the handle's back
            det.forward_device(x.data_ptr(), 8, 320, 320, got.data_ptr())
            rec.classify_device(crops.data_ptr(), 2000, 0, lab.data_ptr(), pr.data_ptr())
        det.close()   # no synchronize
        rec.close()
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(lab, want_lab) and torch.equal(pr, want_pr)

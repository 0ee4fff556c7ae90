"""CTC greedy decode - an EXTENSION without a reference counterpart (the stage BASELINE.json's north_star / configs[2] name; the reference's
recogniser classifies single glyphs, /root/reference/src/char_recognition/model.rs:27-39).  The oracle (oracle/ctc_oracle.py, exact
integers) is pinned by hand-written vectors; the kernel (ocr-rs_amd/csrc/ctc.hip, one wave per crop) is held to the oracle bit for bit
through the C ABI at BASELINE configs[2]'s shape - N = 256 crops, T = 32 columns (a 32 x 128 crop at stride 4), C = 63 classes (the
alphabet of /root/reference/src/utils.rs:7-9 plus the blank) - and at the edges: all-blank crops, all-equal columns, tie columns, T = 1,
T beyond one wave's 64 columns, either end of the class range as the blank."""
import numpy as np
import pytest

from oracle import ctc_oracle as CT


def _onehot(seq, c, blank_fill=0.0):
    x = np.full((1, len(seq), c), blank_fill, np.float32)
    for t, k in enumerate(seq):
        x[0, t, k] = 1.0
    return x


def test_oracle_hand_vectors():
    B = 0
    lab, ln = CT.ctc_greedy_decode(_onehot([B, 3, 3, B, 3, 5, 5, 5, B, B, 7], 9), B)
    assert ln.tolist() == [4] and lab[0, :4].tolist() == [3, 3, 5, 7] and (lab[0, 4:] == -1).all()     # "a a _ a" keeps both a's
    lab, ln = CT.ctc_greedy_decode(_onehot([2, 2, 2, 2], 4), 0)
    assert ln.tolist() == [1] and lab[0, 0] == 2
    lab, ln = CT.ctc_greedy_decode(_onehot([3, 3, 3], 4), 3)                                             # all blank (blank = last class)
    assert ln.tolist() == [0] and (lab == -1).all()
    lab, ln = CT.ctc_greedy_decode(np.zeros((1, 5, 6), np.float32), 5)                                    # all-equal columns: class 0 wins every tie
    assert ln.tolist() == [1] and lab[0, 0] == 0
    lab, ln = CT.ctc_greedy_decode(np.zeros((1, 5, 6), np.float32), 0)                                    # ... which is the blank here
    assert ln.tolist() == [0]
    x = np.zeros((1, 3, 5), np.float32)
    x[0, 0, [2, 4]] = 1.0
    x[0, 1, [4, 2]] = 1.0
    x[0, 2, 1] = -0.0                                                                                     # -0.0 == 0.0: class 0 is first
    lab, ln = CT.ctc_greedy_decode(x, 0)
    assert ln.tolist() == [1] and lab[0, 0] == 2                                                          # ties: the lowest class, twice -> one label


def _cases():
    rng = np.random.default_rng(63)
    out = []
    x = rng.standard_normal((256, 32, 63)).astype(np.float32)                                 # BASELINE configs[2]'s shape
    x[:, :, 62] += 1.5                                                                        # a blank-heavy head, as trained CTC heads are
    x[3] = 0.0                                                                                # all-equal columns
    x[4, :, 62] = 9.0                                                                         # all blank
    x[5, :, :] = -1.0; x[5, :, 17] = 2.0                                                      # one class throughout
    x[6] = -1.0; x[6, ::2, 10] = 5.0; x[6, 1::2, 62] = 5.0                                                 # the same class between blanks: 16 labels
    x[7] = -1.0; x[7, :, 20] = 4.0; x[7, :, 40] = 4.0                                                      # a tie in every column
    out.append(("256x32x63", x, 62))
    out.append(("blank 0", rng.standard_normal((100, 32, 63)).astype(np.float32), 0))
    out.append(("T=1", rng.standard_normal((70, 1, 63)).astype(np.float32), 62))
    out.append(("T=200: chunks of 64 with a carried class", np.repeat(rng.standard_normal((9, 50, 11)).astype(np.float32), 4, axis=1), 10))
    out.append(("T=65", np.repeat(rng.standard_normal((5, 13, 7)).astype(np.float32), 5, axis=1), 3))
    out.append(("C=1: only the blank", np.zeros((3, 8, 1), np.float32), 0))
    out.append(("one crop", rng.standard_normal((1, 32, 63)).astype(np.float32), 31))
    return out


@pytest.mark.gpu
def test_device_ctc_greedy_decode_equals_the_oracle():
    import ocr_rs_amd  # noqa: F401
    from ocr_rs_amd import capi
    from ocr_rs_amd import weights as W
    rec = capi.Recognizer(W.pack_blob(W.make_rec_weights(0)), 0)
    for name, x, blank in _cases():
        want_l, want_n = CT.ctc_greedy_decode(x, blank)
        got_l, got_n = rec.ctc_greedy_decode(x, blank)
        assert np.array_equal(got_n, want_n), name
        assert np.array_equal(got_l, want_l), name
    # device memory
    import torch
    name, x, blank = _cases()[0]
    xd = torch.from_numpy(x).cuda()
    lab = torch.empty((x.shape[0], x.shape[1]), dtype=torch.int32, device="cuda")
    ln = torch.empty(x.shape[0], dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rec.ctc_greedy_decode_device(xd.data_ptr(), x.shape[0], x.shape[1], x.shape[2], blank, lab.data_ptr(), ln.data_ptr())
    want_l, want_n = CT.ctc_greedy_decode(x, blank)
    assert np.array_equal(lab.cpu().numpy(), want_l) and np.array_equal(ln.cpu().numpy(), want_n)
    assert int(want_n[6]) == 16 and int(want_n[4]) == 0 and int(want_n[3]) == 1 and int(want_n[7]) == 1
    with pytest.raises(capi.OcrError):
        rec.ctc_greedy_decode(x, 63)
    rec.close()


@pytest.mark.gpu
def test_reference_named_mirror_decodes_to_strings():
    """char_recognition.Net.ctc_greedy_decode: classes 0..61 are the reference's alphabet (utils.rs:7), class 62 the blank"""
    import ocr_rs_amd  # noqa: F401
    from ocr_rs_amd import char_recognition as cr
    from ocr_rs_amd import weights as W
    net = cr.Net(W.pack_blob(W.make_rec_weights(0)), 0)
    word = [7, 7, 62, 30, 62, 37, 37, 62, 37, 40, 62, 62, 52, 53]        # H e l l o 0 1 with repeats and blanks
    x = np.full((2, len(word), 63), -4.0, np.float32)
    for t, k in enumerate(word):
        x[0, t, k] = 3.0
    x[1, :, 62] = 1.0                                                     # all blank
    assert net.ctc_greedy_decode(x) == ["Hello01", ""]
    net.close()

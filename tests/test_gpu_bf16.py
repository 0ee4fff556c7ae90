"""OCR_PRECISION_BF16 (BASELINE config 5): the opt-in reduced precision of the detector.

The reference is f32 only, so there is no reference result to match here.  Bit-level parity of the bf16
arithmetic is established per kernel (tests/test_gpu_conv_kernel.py: same rounded operands, at most one
bf16 ulp).  End to end the stack of ~30 bf16 roundings is chaotic - oracle/torch_ref.det_forward_bf16
itself moves by ~9e-3 when the input frame changes by 1e-6 relative - so the whole-network bars are
statistical:
  * the distance of the HIP bf16 map from the f32 map is no larger than what the restated bf16 arithmetic
    (det_forward_bf16 on ATen-CPU) loses against f32 on the same frames (x DRIFT_FACTOR), max and mean;
  * switching back to f32 restores the f32 result bit for bit (the parity configuration is untouched).
"""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import weights as W
from oracle import torch_ref as T

pytestmark = pytest.mark.gpu

DRIFT_FACTOR = 1.5   # HIP bf16-vs-f32 distance may exceed the restatement's by this factor (both are noise-like)


@pytest.fixture(scope="module")
def det_w():
    return W.make_det_weights(0)


@pytest.fixture()
def det(det_w):
    d = capi.Detector(W.pack_blob(det_w), 0)
    yield d
    d.close()


@pytest.mark.parametrize("n,h,w,seed", [(2, 64, 96, 7), (1, 32, 32, 3), (3, 96, 160, 4), (1, 320, 320, 9)])
def test_bf16_drift_matches_restated_arithmetic(det, det_w, n, h, w, seed):
    x = W.synth_image_batch(seed, n, h, w)
    f32 = det.forward_host(x)
    det.set_precision(capi.PRECISION_BF16)
    got = det.forward_host(x)
    ref32 = T.det_forward(det_w, x)
    ref16 = T.det_forward_bf16(det_w, x)
    d_hip, d_ref = np.abs(got - f32), np.abs(ref16 - ref32)
    print(f"bf16 {n}x{h}x{w}: HIP max {d_hip.max():.3e} mean {d_hip.mean():.3e} | restated max {d_ref.max():.3e} "
          f"mean {d_ref.mean():.3e} | HIP vs restated max {np.abs(got - ref16).max():.3e}")
    assert got.shape == (n, 1, h, w) and np.isfinite(got).all()
    assert d_hip.max() > 0.0   # the bf16 kernels really ran
    assert d_hip.max() <= DRIFT_FACTOR * d_ref.max()
    assert d_hip.mean() <= DRIFT_FACTOR * d_ref.mean()
    assert np.abs(got - ref16).max() <= 2.0 * d_ref.max()


def test_precision_switch_restores_f32_bits(det):
    x = W.synth_image_batch(11, 2, 96, 96)
    a = det.forward_host(x)
    det.set_precision(capi.PRECISION_BF16)
    b = det.forward_host(x)
    det.set_precision(capi.PRECISION_F32)
    c = det.forward_host(x)
    assert np.array_equal(a, c)
    assert not np.array_equal(a, b)


def test_bad_precision_is_invalid(det):
    with pytest.raises(capi.OcrError) as e:
        det.set_precision(7)
    assert e.value.code == 1


def test_bf16_postprocess_runs_on_f32_map(det):
    """The boundary after the network is unchanged: the f32 probability map feeds the same post-processing."""
    x = W.synth_image_batch(5, 2, 128, 128)
    det.set_precision(capi.PRECISION_BF16)
    prob = det.forward_host(x)
    polys, scores = det.postprocess(prob, 2, 128, 128, np.ones((2, 2)), capi.MEM_HOST, capi.default_params(True))
    assert len(polys) == 2 and len(scores) == 2


def test_bf16_bin_conv1_forms_meet_the_same_bars(det_w):
    """bin_conv1 over the pyramid in bf16: p2's 3x3 term as the patch-staged 64 -> 64 conv on top of the phase launch (default)
    or inside the phase launch (`pyr_p2_direct=0`).  One more bf16 rounding of a partial sum: both forms sit within the drift of
    the restated bf16 arithmetic, and within it of each other."""
    x = W.synth_image_batch(21, 2, 96, 160)
    ref32, ref16 = T.det_forward(det_w, x), T.det_forward_bf16(det_w, x)
    d_ref = np.abs(ref16 - ref32)
    maps = []
    for opt in ("precision=bf16", "precision=bf16;pyr_p2_direct=0", "precision=bf16;overlap=0"):   # (the default schedule is overlap=3)
        d = capi.Detector(W.pack_blob(det_w), 0, options=opt)
        try:
            maps.append(d.forward_host(x))
        finally:
            d.close()
        assert np.abs(maps[-1] - ref32).max() <= DRIFT_FACTOR * d_ref.max(), opt
        assert np.abs(maps[-1] - ref32).mean() <= DRIFT_FACTOR * d_ref.mean(), opt
    assert not np.array_equal(maps[0], maps[1])   # two different schedules really ran
    assert np.abs(maps[0] - maps[1]).max() <= DRIFT_FACTOR * d_ref.max()
    # the side-stream schedule (bin_conv1's p2 term rounded to bf16 BEFORE the pyramid's sum is added instead of after) against the one-stream one
    assert np.abs(maps[0] - maps[2]).max() <= DRIFT_FACTOR * d_ref.max()


@pytest.mark.parametrize("n,h,w", [(2, 64, 96), (1, 320, 320), (3, 32, 32)])
def test_bf16_fused_basic_blocks_give_the_same_map_bit_for_bit(det_w, n, h, w):
    """bf16_block_fuse=1 (default: each BasicBlock of layer1 one launch, basic_block_bf16_c64.hip) against =0 (two conv3x3_bf16_c64
    launches per block): identical probability maps, to the last bit (model.rs:40-55)"""
    x = W.synth_image_batch(21, n, h, w)
    maps = []
    for opt in ("", "bf16_block_fuse=0"):
        d = capi.Detector(W.pack_blob(det_w), 0, options=opt)
        d.set_precision(capi.PRECISION_BF16)
        maps.append(d.forward_host(x))
        d.close()
    assert np.array_equal(maps[0], maps[1])

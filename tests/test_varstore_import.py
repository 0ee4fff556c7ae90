"""weight import from a tch VarStore file (`vs.load(file)`, text_detection/mod.rs:41-44).
The reference ships no weight file, so the importer is pinned to an archive written by the libtorch C++
calls that tch's VarStore::save reaches (tests/golden/make_varstore_fixture.cpp -> varstore_small.ot)."""
import os

import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import weights as W

FIXTURE = {   # name -> (shape, scale): value[i] = i * scale + 1, see make_varstore_fixture.cpp
    "conv1.weight": ((4, 1, 5, 5), 0.01), "conv1.bias": ((4,), 0.1),
    "fc2.weight": ((3, 8), -0.02), "fc2.bias": ((3,), 1.0),
    "layer1.0.bn1.running_mean": ((4,), 0.25), "layer1.0.bn1.running_var": ((4,), 2.0),
    "layer2.0.downsample.1.weight": ((2,), 3.0),
}


def test_varstore_archive_is_read_by_name(golden_dir):
    got = W.load_varstore(os.path.join(golden_dir, "varstore_small.ot"))
    assert set(got) == set(FIXTURE)
    for name, (shape, scale) in FIXTURE.items():
        want = (np.arange(int(np.prod(shape)), dtype=np.float32) * np.float32(scale) + np.float32(1.0)).reshape(shape)
        assert got[name].dtype == np.float32 and got[name].shape == shape
        assert np.array_equal(got[name], want), name


def test_varstore_kind_check_names_what_is_wrong(golden_dir):
    with pytest.raises(ValueError) as e:
        W.load_varstore(os.path.join(golden_dir, "varstore_small.ot"), kind="rec")
    msg = str(e.value)
    assert "conv1.weight: shape (4, 1, 5, 5), expected (32, 1, 5, 5)" in msg and "missing conv2.weight" in msg


def test_imported_tensors_pack_into_the_blob_the_abi_takes(golden_dir):
    got = W.load_varstore(os.path.join(golden_dir, "varstore_small.ot"))
    back = W.unpack_blob(W.pack_blob(got))
    assert set(back) == set(got) and all(np.array_equal(back[k], got[k]) for k in got)


def test_recogniser_names_as_tch_writes_them():
    """Net::new puts all four layers on one nn::Path (char_recognition/model.rs:13-24): tch de-duplicates the
    colliding names with __N suffixes.  Both registration orders map onto conv1 / conv2 / fc1 / fc2 by shape."""
    want = W.make_rec_weights(3)
    weight_first = ["weight", "bias", "weight__2", "bias__3", "weight__4", "bias__5", "weight__6", "bias__7"]
    bias_first_convs = ["bias", "weight", "bias__2", "weight__3", "weight__4", "bias__5", "weight__6", "bias__7"]
    for names in (weight_first, bias_first_convs):
        leafs = {}
        for (name, _), layer in zip(W.rec_param_specs(), [0, 0, 1, 1, 2, 2, 3, 3]):
            leaf = name.split(".")[1]
            tch_name = next(n for n in names[2 * layer:2 * layer + 2] if n.split("__")[0] == leaf)
            leafs[tch_name] = want[name]
        got = W.rename_tch_rec(leafs)
        assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want)
    assert W.rename_tch_rec(want) is want          # this library's own dotted names pass through

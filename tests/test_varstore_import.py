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

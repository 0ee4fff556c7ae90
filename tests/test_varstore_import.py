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


# ---------------------------------------------------------------- the library's own reader (C++, no torch)
from ocr_rs_amd import capi  # noqa: E402


def test_native_reader_matches_the_torch_reader_on_the_fixture(golden_dir):
    path = os.path.join(golden_dir, "varstore_small.ot")
    got = W.unpack_blob(capi.varstore_to_blob(path))
    ref = W.load_varstore(path)
    assert list(got) == list(ref)                       # same tensors, archive order
    assert all(got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]) for k in ref)


def test_native_reader_rejects_what_is_not_an_archive(tmp_path):
    bad = tmp_path / "bad.ot"
    bad.write_bytes(b"not a zip at all" * 10)
    with pytest.raises(capi.OcrError) as e:
        capi.varstore_to_blob(str(bad))
    assert e.value.code == 2 and "zip" in str(e.value)
    with pytest.raises(capi.OcrError):
        capi.varstore_to_blob(str(tmp_path / "missing.ot"))


@pytest.fixture(scope="module")
def mkvs(tmp_path_factory):
    """The archive writer, compiled against this image's libtorch (the calls torch-sys' at_save_multi makes)."""
    import subprocess
    import torch
    t = os.path.dirname(torch.__file__)
    exe = str(tmp_path_factory.mktemp("mkvs") / "mkvs")
    src = os.path.join(os.path.dirname(__file__), "golden", "make_varstore_fixture.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-D_GLIBCXX_USE_CXX11_ABI=1", f"-I{t}/include", f"-I{t}/include/torch/csrc/api/include",
           src, "-o", exe, f"-L{t}/lib", "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{t}/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the libtorch archive writer here: " + r.stderr[-300:])
    return exe


def _write_archive(mkvs, path, named_shapes):
    import subprocess
    spec = path + ".spec"
    with open(spec, "w") as f:
        for name, shape in named_shapes:
            f.write(name + " " + " ".join(str(d) for d in shape) + "\n")
    subprocess.run([mkvs, path, spec], check=True)
    out = {}
    for t, (name, shape) in enumerate(named_shapes):
        n = int(np.prod(shape))
        out[name] = (((np.arange(n) % 251) - 125) / 128.0 + t).astype(np.float32).reshape(shape)
    return out


def test_native_reader_detector_varstore_at_full_size(mkvs, tmp_path):
    """All 121 tensors / 12 180 097 parameters of resnet18() (model.rs:68-105) under their VarStore names."""
    path = str(tmp_path / "text_detection.model")
    want = _write_archive(mkvs, path, W.det_param_specs())
    got = W.unpack_blob(capi.varstore_to_blob(path, capi.VARSTORE_DET))
    assert set(got) == set(want) and sum(v.size for v in got.values()) == 12180097
    assert all(got[k].shape == want[k].shape and np.array_equal(got[k], want[k]) for k in want)


@pytest.mark.parametrize("order", ["weight_first", "bias_first_in_convs"])
def test_native_reader_recogniser_names_as_tch_writes_them(mkvs, tmp_path, order):
    specs = dict(W.rec_param_specs())
    if order == "weight_first":
        names = ["weight", "bias", "weight__2", "bias__3", "weight__4", "bias__5", "weight__6", "bias__7"]
        layers = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    else:
        names = ["bias", "weight", "bias__2", "weight__3", "weight__4", "bias__5", "weight__6", "bias__7"]
        layers = ["conv1.bias", "conv1.weight", "conv2.bias", "conv2.weight", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    path = str(tmp_path / "char_rec_conv_net.model")
    stored = _write_archive(mkvs, path, [(n, specs[l]) for n, l in zip(names, layers)])
    got = W.unpack_blob(capi.varstore_to_blob(path, capi.VARSTORE_REC))
    assert set(got) == set(specs)
    for n, l in zip(names, layers):
        assert np.array_equal(got[l], stored[n]), l
    # and the torch-based importer agrees
    ref = W.load_varstore(path, kind="rec")
    assert all(np.array_equal(ref[k], got[k]) for k in specs)


@pytest.mark.gpu
def test_create_from_varstore_is_vs_load(mkvs, tmp_path):
    """`vs.load(file)` in one call (text_detection/mod.rs:41-44, char_recognition/mod.rs:46): the engines built
    straight from tch archives behave exactly like the ones built from the blob of the same tensors."""
    specs = dict(W.rec_param_specs())
    names = ["weight", "bias", "weight__2", "bias__3", "weight__4", "bias__5", "weight__6", "bias__7"]
    layers = [n for n, _ in W.rec_param_specs()]
    rpath = str(tmp_path / "char_rec_conv_net.model")
    _write_archive(mkvs, rpath, [(n, specs[l]) for n, l in zip(names, layers)])
    crops = W.synth_crops(3, 37)
    a = capi.Recognizer(None, 0, varstore_path=rpath)
    b = capi.Recognizer(capi.varstore_to_blob(rpath, capi.VARSTORE_REC), 0)
    la, lb = a.forward_host(crops), b.forward_host(crops)
    assert np.isfinite(la).all() and np.array_equal(la, lb)
    a.close()
    b.close()
    dpath = str(tmp_path / "text_detection.model")
    _write_archive(mkvs, dpath, W.det_param_specs())
    x = W.synth_image_batch(2, 1, 64, 64)
    d1 = capi.Detector(None, 0, varstore_path=dpath)
    d2 = capi.Detector(capi.varstore_to_blob(dpath, capi.VARSTORE_DET), 0)
    assert np.array_equal(d1.forward_host(x), d2.forward_host(x), equal_nan=True)
    d1.close()
    d2.close()
    with pytest.raises(capi.OcrError) as e:      # a recogniser file is not a detector: named, not a crash
        capi.Detector(None, 0, varstore_path=rpath)
    assert e.value.code == 2 and "conv1.weight" in str(e.value)

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


# ---------------------------------------------------------------- archives in libtorch 1.7.0's container layout, built byte by byte
# /root/reference/src/utils.rs:55-63 saves through tch 0.3.0 -> libtorch 1.7.0.  That writer is not available here; what it emits
# differs from this image's libtorch 2.10 in container details, restated below from the published 1.7 sources
# (caffe2/serialize/inline_container.cc, torch/csrc/jit/serialization/pickler.cpp, export_module.cpp):
#   * every record under "<file stem>/"; order data/0 .. data/N-1, data.pkl, code/..., constants.pkl, version
#   * version record "3\n"; no byteorder, no .data/serialization_id records
#   * records STORED, their payload aligned to 64 bytes by an "FB" extra field padded with 'Z'
#   * protocol-2 pickle of the module object: GLOBAL __torch__ Module, EMPTY_TUPLE, NEWOBJ, EMPTY_DICT, MARK, (BINUNICODE name,
#     tensor)*, SETITEMS, BUILD; a tensor = GLOBAL torch._utils _rebuild_tensor_v2, MARK, [MARK 'storage' GLOBAL torch FloatStorage
#     key 'cpu' numel TUPLE BINPERSID], offset, MARK sizes TUPLE, MARK strides TUPLE, requires_grad, OrderedDict() , TUPLE, REDUCE;
#     tuples always MARK .. TUPLE (no TUPLE1-3), every global and string memoised with BINPUT / LONG_BINPUT
import struct  # noqa: E402
import zlib  # noqa: E402


class _Pickle17:
    def __init__(self):
        self.b = bytearray(b"\x80\x02")
        self.memo = {}
        self.next = 0

    def _put(self, key):
        self.memo[key] = self.next
        self.b += (b"q" + bytes([self.next])) if self.next < 256 else (b"r" + struct.pack("<I", self.next))
        self.next += 1

    def _get(self, key):
        k = self.memo[key]
        self.b += (b"h" + bytes([k])) if k < 256 else (b"j" + struct.pack("<I", k))

    def glob(self, module, name):
        key = ("g", module, name)
        if key in self.memo:
            return self._get(key)
        self.b += b"c" + module.encode() + b"\n" + name.encode() + b"\n"
        self._put(key)

    def string(self, s):
        key = ("s", s)
        if key in self.memo:
            return self._get(key)
        e = s.encode()
        self.b += b"X" + struct.pack("<I", len(e)) + e
        self._put(key)

    def integer(self, v):
        if 0 <= v < 256:
            self.b += b"K" + bytes([v])
        elif 0 <= v < 65536:
            self.b += b"M" + struct.pack("<H", v)
        elif -2 ** 31 <= v < 2 ** 31:
            self.b += b"J" + struct.pack("<i", v)
        else:
            self.b += b"\x8a\x08" + struct.pack("<q", v)

    def int_tuple(self, vals):
        self.b += b"("
        for v in vals:
            self.integer(int(v))
        self.b += b"t"

    def tensor(self, key, numel, offset, sizes, strides):
        self.glob("torch._utils", "_rebuild_tensor_v2")
        self.b += b"("
        self.b += b"("
        self.string("storage")
        self.glob("torch", "FloatStorage")
        self.string(str(key))
        self.string("cpu")
        self.integer(numel)
        self.b += b"tQ"
        self.integer(offset)
        self.int_tuple(sizes)
        self.int_tuple(strides)
        self.b += b"\x89"                      # requires_grad = False
        self.glob("collections", "OrderedDict")
        self.b += b")R"                        # backward hooks: OrderedDict()
        self.b += b"tR"


def _contig(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return tuple(reversed(st))


def _archive_1_7(path, tensors, deflate=None):
    """tensors: [(name, storage f32 array (flat), offset, sizes, strides)]; deflate: record name (without the stem) to compress."""
    stem = os.path.splitext(os.path.basename(path))[0]
    pk = _Pickle17()
    pk.glob("__torch__", "Module")
    pk.b += b")\x81}("
    for k, (name, storage, offset, sizes, strides) in enumerate(tensors):
        pk.string(name)
        pk.tensor(k, storage.size, offset, sizes, strides)
    pk.b += b"ub."
    records = [(f"data/{k}", t[1].astype("<f4").tobytes()) for k, t in enumerate(tensors)]
    records += [("data.pkl", bytes(pk.b)),
                ("code/__torch__.py", b"class Module(Module):\n  __parameters__ = [" + b", ".join(b'"' + t[0].encode() + b'"' for t in tensors) + b"]\n"),
                ("code/__torch__.py.debug_pkl", b"\x80\x02).")]
    records += [("constants.pkl", b"\x80\x02).")]
    records += [("version", b"3\n")]
    out, central = bytearray(), bytearray()
    for name, data in records:
        full = (stem + "/" + name).encode()
        method, payload = 0, data
        if deflate == name:
            c = zlib.compressobj(9, zlib.DEFLATED, -15)
            method, payload = 8, c.compress(data) + c.flush()
        start = len(out) + 30 + len(full) + 4                    # where the payload would begin behind a bare "FB" extra field
        pad = (64 - start % 64) % 64
        extra = b"FB" + struct.pack("<H", pad) + b"Z" * pad
        crc = zlib.crc32(data) & 0xFFFFFFFF
        off = len(out)
        out += struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, method, 0, 0, crc, len(payload), len(data), len(full), len(extra)) + full + extra
        assert method or len(out) % 64 == 0
        out += payload
        central += struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 20, 20, 0, method, 0, 0, crc, len(payload), len(data), len(full), 0, 0, 0, 0, 0, off) + full
    cd_off = len(out)
    out += central
    out += struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(records), len(records), len(central), cd_off, 0)
    with open(path, "wb") as f:
        f.write(out)


def _values(t, n):
    return ((np.arange(n) % 251) - 125).astype(np.float32) / np.float32(128.0) + np.float32(t)


def test_native_reader_libtorch_1_7_detector_archive(tmp_path):
    """The detector's 121 tensors under their VarStore names in a libtorch-1.7-layout container (more than 256 memo entries:
    LONG_BINPUT / LONG_BINGET in the pickle)."""
    path = str(tmp_path / "text_detection.model")
    want, tensors = {}, []
    for t, (name, shape) in enumerate(W.det_param_specs()):
        v = _values(t, int(np.prod(shape)))
        want[name] = v.reshape(shape)
        tensors.append((name, v, 0, shape, _contig(shape)))
    _archive_1_7(path, tensors)
    got = W.unpack_blob(capi.varstore_to_blob(path, capi.VARSTORE_DET))
    assert set(got) == set(want) and sum(v.size for v in got.values()) == 12180097
    assert all(got[k].shape == want[k].shape and np.array_equal(got[k], want[k]) for k in want)


def test_native_reader_libtorch_1_7_recogniser_archive_with_views(tmp_path):
    """The recogniser under tch's colliding leaf names, in the 1.7 layout - with what an archive may legitimately hold beyond
    contiguous tensors: a storage offset and a transposed (non-contiguous) fc weight.  Strides are resolved, not assumed."""
    specs = dict(W.rec_param_specs())
    names = ["weight", "bias", "weight__2", "bias__3", "weight__4", "bias__5", "weight__6", "bias__7"]
    layers = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    path = str(tmp_path / "char_rec_conv_net.model")
    want, tensors = {}, []
    for t, (n, l) in enumerate(zip(names, layers)):
        shape = specs[l]
        cnt = int(np.prod(shape))
        if l == "fc1.weight":       # stored transposed: [1024][512] in memory, seen as [512][1024] through strides (1, 512)
            st = _values(t, cnt)
            want[l] = st.reshape(shape[1], shape[0]).T.copy()
            tensors.append((n, st, 0, shape, (1, shape[0])))
        elif l == "conv2.bias":     # a view 7 elements into a larger storage
            st = _values(t, cnt + 11)
            want[l] = st[7:7 + cnt].reshape(shape)
            tensors.append((n, st, 7, shape, _contig(shape)))
        else:
            st = _values(t, cnt)
            want[l] = st.reshape(shape)
            tensors.append((n, st, 0, shape, _contig(shape)))
    _archive_1_7(path, tensors)
    got = W.unpack_blob(capi.varstore_to_blob(path, capi.VARSTORE_REC))
    assert set(got) == set(specs)
    assert all(got[k].shape == tuple(specs[k]) and np.array_equal(got[k], want[k]) for k in specs)


def test_native_reader_names_a_compressed_record_and_a_short_storage(tmp_path):
    specs = W.rec_param_specs()
    names = ["weight", "bias", "weight__2", "bias__3", "weight__4", "bias__5", "weight__6", "bias__7"]
    tensors = [(n, _values(t, int(np.prod(s))), 0, s, _contig(s)) for t, (n, (_, s)) in enumerate(zip(names, specs))]
    p1 = str(tmp_path / "deflated.model")
    _archive_1_7(p1, tensors, deflate="data/2")
    with pytest.raises(capi.OcrError) as e:
        capi.varstore_to_blob(p1, capi.VARSTORE_REC)
    assert e.value.code == 2 and "compressed" in str(e.value) and "data/2" in str(e.value)
    p2 = str(tmp_path / "short.model")
    bad = list(tensors)
    bad[4] = (bad[4][0], bad[4][1][:1000], 0, bad[4][3], bad[4][4])      # fc1.weight's storage cut short
    _archive_1_7(p2, bad)
    with pytest.raises(capi.OcrError) as e:
        capi.varstore_to_blob(p2, capi.VARSTORE_REC)
    assert e.value.code == 2 and "past its storage" in str(e.value)

"""ctypes binding of include/ocr_amd.h (the same C ABI the Rust shim of
INTEGRATION.md binds).  No torch types cross this boundary: device tensors are
passed as raw pointers (`tensor.data_ptr()`).

The product has NO CPU fallback: every compute entry point needs libocr_amd.so
and a gfx950 device and raises OcrError otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCR_AMD_LIB") or os.path.join(_HERE, "lib", "libocr_amd.so")

MEM_HOST, MEM_DEVICE = 0, 1
ELEM_F32, ELEM_U8 = 0, 1
PRECISION_F32, PRECISION_BF16 = 0, 1

EXPORTS = [
    "ocr_last_error", "ocr_version", "ocr_device_count",
    "ocr_varstore_to_blob", "ocr_blob_free", "ocr_det_create_from_varstore", "ocr_rec_create_from_varstore",
    "ocr_det_create", "ocr_det_create_with_options", "ocr_det_destroy", "ocr_det_set_stream", "ocr_det_set_precision", "ocr_det_forward",
    "ocr_det_forward_u8", "ocr_host_alloc", "ocr_host_free", "ocr_det_detect_pipelined_host",
    "ocr_det_forward_async", "ocr_det_synchronize", "ocr_det_forward_profile",
    "ocr_preprocess_image", "ocr_postproc_default_params", "ocr_det_postprocess", "ocr_det_post_stats", "ocr_det_detect_pipelined", "ocr_polygons_free",
    "ocr_extract_crops", "ocr_evaluate_image", "ocr_combine_results",
    "ocr_rec_create", "ocr_rec_destroy", "ocr_rec_set_stream", "ocr_rec_set_options", "ocr_rec_synchronize",
    "ocr_rec_forward", "ocr_rec_classify_async", "ocr_rec_classify_profile", "ocr_rec_classify", "ocr_rec_alphabet", "ocr_ctc_greedy_decode",
    "ocr_comm_unique_id", "ocr_comm_rccl_version", "ocr_comm_create", "ocr_comm_destroy",
    "ocr_comm_all_gather_polygons", "ocr_comm_all_gather_labels",
]


class OcrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ocr_amd error {code}: {msg}")
        self.code = code


class PostprocParams(C.Structure):
    _fields_ = [("thresh", C.c_double), ("box_thresh", C.c_double), ("min_size", C.c_double),
                ("unclip_ratio", C.c_double), ("skip_degenerate", C.c_int32), ("reserved", C.c_int32)]


class MetricsItem(C.Structure):
    _fields_ = [("precision", C.c_double), ("recall", C.c_double), ("hmean", C.c_double),
                ("gt_care", C.c_int32), ("det_care", C.c_int32), ("det_matched", C.c_int32)]


class Polygons(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("n_polygons", C.c_int32), ("n_vertices", C.c_int32),
                ("img_offsets", C.POINTER(C.c_int32)), ("poly_offsets", C.POINTER(C.c_int32)),
                ("xy", C.POINTER(C.c_uint32)), ("scores", C.POINTER(C.c_double))]


_lib = None
_hip_shared = False


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7) and libhsa-runtime64.so and
    link to them by FILE name; libocr_amd.so needs `libamdhip64.so.7`.  Loaded after torch, the library binds to torch's copy (the SONAME
    matches) and the process has one runtime.  Loaded BEFORE torch, it brings /opt/rocm's copy in, torch then loads its own beside it (its
    NEEDED name matches no loaded SONAME) and the second runtime finds the device taken: torch's lazy init fails with "No HIP GPUs are
    available".  So this harness loads torch's copy first when a torch installation exists and has not been imported yet - without
    importing torch.  (A host that does not use torch - the Rust / C++ caller of INTEGRATION.md - has one runtime anyway.)"""
    global _hip_shared
    if _hip_shared:
        return
    _hip_shared = True
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:   # no torch, or an unusual layout: nothing to share
        pass


def lib() -> C.CDLL:
    """Loads libocr_amd.so; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcrError(-1, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.ocr_last_error.restype = C.c_char_p
        L.ocr_version.restype = C.c_char_p
        L.ocr_rec_alphabet.restype = C.c_char_p
        L.ocr_det_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_det_destroy.argtypes = [C.c_void_p]
        L.ocr_det_destroy.restype = None
        L.ocr_det_create_with_options.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ocr_det_create_from_varstore.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_rec_create_from_varstore.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_varstore_to_blob.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ocr_blob_free.argtypes = [C.c_void_p]
        L.ocr_blob_free.restype = None
        L.ocr_det_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ocr_det_set_precision.argtypes = [C.c_void_p, C.c_int]
        L.ocr_det_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ocr_det_forward_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ocr_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.ocr_host_free.argtypes = [C.c_void_p]
        L.ocr_host_free.restype = None
        L.ocr_det_detect_pipelined_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                    C.POINTER(C.c_double), C.POINTER(PostprocParams),
                                                    C.POINTER(C.POINTER(Polygons))]
        L.ocr_det_forward_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_float]
        L.ocr_det_synchronize.argtypes = [C.c_void_p]
        L.ocr_det_forward_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ocr_preprocess_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.ocr_postproc_default_params.argtypes = [C.POINTER(PostprocParams)]
        L.ocr_postproc_default_params.restype = None
        L.ocr_det_post_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ocr_ctc_greedy_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ocr_det_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(PostprocParams),
                                          C.POINTER(C.POINTER(Polygons))]
        L.ocr_det_detect_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.POINTER(C.c_double), C.POINTER(PostprocParams),
                                               C.POINTER(C.POINTER(Polygons))]
        L.ocr_polygons_free.argtypes = [C.POINTER(Polygons)]
        L.ocr_polygons_free.restype = None
        L.ocr_evaluate_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(MetricsItem)]
        L.ocr_combine_results.argtypes = [C.POINTER(MetricsItem), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]
        L.ocr_extract_crops.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Polygons),
                                        C.POINTER(C.c_double), C.c_void_p]
        L.ocr_rec_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_rec_destroy.argtypes = [C.c_void_p]
        L.ocr_rec_destroy.restype = None
        L.ocr_rec_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ocr_rec_synchronize.argtypes = [C.c_void_p]
        L.ocr_rec_set_options.argtypes = [C.c_void_p, C.c_char_p]
        L.ocr_rec_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ocr_rec_classify_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ocr_rec_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ocr_rec_classify_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ocr_comm_unique_id.argtypes = [C.c_void_p]
        L.ocr_comm_rccl_version.argtypes = [C.POINTER(C.c_int)]
        L.ocr_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_comm_destroy.argtypes = [C.c_void_p]
        L.ocr_comm_destroy.restype = None
        L.ocr_comm_all_gather_polygons.argtypes = [C.c_void_p, C.POINTER(Polygons), C.POINTER(C.POINTER(Polygons))]
        L.ocr_comm_all_gather_labels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.POINTER(C.c_int)]
        _lib = L
    return _lib


_test_lib = None
TEST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libocr_amd_test.so")


def test_lib() -> C.CDLL:
    """libocr_amd_test.so: the ocr_test_* hooks (kernel-level parity, host geometry, tuning aids).  A separate
    library built from the same objects plus test_hooks.o - nothing of it ships in libocr_amd.so, whose internals are
    not exported.  Hooks take the handles the product library created (same classes, same process)."""
    global _test_lib
    if _test_lib is None:
        lib()
        if not os.path.exists(TEST_LIB_PATH):
            raise OcrError(-1, f"{TEST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _share_torch_hip_runtime()
        L = C.CDLL(TEST_LIB_PATH)
        L.ocr_test_contour_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_int, C.POINTER(C.c_int)]
        L.ocr_test_expand_polygon.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                              C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ocr_test_min_area_box.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        L.ocr_test_box_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p]
        L.ocr_test_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.POINTER(C.c_float)]
        L.ocr_test_conv_run.argtypes = ([C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] +
                                        [C.c_int] * 3 + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 2)
        L.ocr_test_set_conv_tile.argtypes = [C.c_int]
        L.ocr_test_bf16_basic_block.argtypes = ([C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int] * 3 +
                                                [C.c_void_p, C.POINTER(C.c_float)])
        L.ocr_test_winograd_conv.argtypes = ([C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int] +
                                             [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p])
        L.ocr_test_det_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ocr_test_comm_assemble.argtypes = [C.POINTER(C.POINTER(Polygons)), C.c_int, C.POINTER(C.POINTER(Polygons))]
        _test_lib = L
    return _test_lib


def check(code: int) -> None:
    if code != 0:
        msg = lib().ocr_last_error().decode()
        if not msg and _test_lib is not None:   # the failure came from a test hook: that library keeps its own message
            _test_lib.ocr_last_error.restype = C.c_char_p
            msg = _test_lib.ocr_last_error().decode()
        raise OcrError(code, msg)


def use_test_library() -> None:
    """Tuning tools whose hooks change library-wide state (tile override, ablation bits, stamp buffers) must run the
    detector from the SAME library the hooks live in: call this before the first lib() - libocr_amd_test.so carries the
    whole C ABI besides the hooks (it is built from the same objects)."""
    global LIB_PATH
    if _lib is not None:
        raise OcrError(-1, "use_test_library() must come before the first library call")
    LIB_PATH = TEST_LIB_PATH


def _ptr(a) -> int:
    """Raw address of a numpy array or of a torch tensor (host or device)."""
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


def polygons_to_python(pp) -> Tuple[List[List[List[Tuple[int, int]]]], List[List[float]]]:
    """CSR block -> PolygonScores{polygons: Vec<MultiPolygon<u32>>, scores: Vec<Vec<f64>>}."""
    p = pp.contents
    ni, npoly, nv = p.n_images, p.n_polygons, p.n_vertices
    img = np.ctypeslib.as_array(p.img_offsets, shape=(ni + 1,)).tolist()
    po = np.ctypeslib.as_array(p.poly_offsets, shape=(npoly + 1,)).tolist() if npoly else [0]
    pts = list(map(tuple, np.ctypeslib.as_array(p.xy, shape=(2 * nv,)).reshape(-1, 2).tolist())) if nv else []
    sc = np.ctypeslib.as_array(p.scores, shape=(npoly,)).tolist() if npoly else []
    polys = [[pts[po[k]:po[k + 1]] for k in range(img[b], img[b + 1])] for b in range(ni)]
    scores = [sc[img[b]:img[b + 1]] for b in range(ni)]
    return polys, scores


VARSTORE_RAW, VARSTORE_DET, VARSTORE_REC = 0, 1, 2


def varstore_to_blob(path: str, kind: int = VARSTORE_RAW) -> bytes:
    """tch VarStore file -> OCRW blob through the library's own zip + pickle reader (host only, no torch)."""
    blob, n = C.c_void_p(), C.c_size_t(0)
    check(lib().ocr_varstore_to_blob(os.fsencode(path), kind, C.byref(blob), C.byref(n)))
    try:
        return C.string_at(blob, n.value)
    finally:
        lib().ocr_blob_free(blob)


def default_params(skip_degenerate: bool = False) -> PostprocParams:
    p = PostprocParams()
    lib().ocr_postproc_default_params(C.byref(p))
    p.skip_degenerate = 1 if skip_degenerate else 0
    return p


class HostBuffer:
    """Pinned host memory from ocr_host_alloc, viewed as a numpy array (frames / maps of the host-memory entry points)."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(lib().ocr_host_alloc(self.nbytes, C.byref(self._p)))
        buf = (C.c_char * self.nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def close(self) -> None:
        if self._p:
            self.array = None
            lib().ocr_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Detector:
    """Owns an ocr_det_t.  Mirrors `resnet18(&vs.root())` + `vs.load(..)`
    (/root/reference/src/text_detection/mod.rs:35-44)."""

    def __init__(self, weights_blob: Optional[bytes], device: int = 0, varstore_path: Optional[str] = None,
                 options: Optional[str] = None):
        """options: "key=value;..." engine options of ocr_det_create_with_options (None = defaults)."""
        self._h = C.c_void_p()
        self._blob = weights_blob
        if varstore_path is not None:   # `vs.load(file)`: the library reads the tch archive itself
            check(lib().ocr_det_create_from_varstore(os.fsencode(varstore_path), device, C.byref(self._h)))
        else:
            check(lib().ocr_det_create_with_options(weights_blob, len(weights_blob), device,
                                                    options.encode() if options else None, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            lib().ocr_det_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, raw_stream: Optional[int]) -> None:
        check(lib().ocr_det_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    def set_precision(self, precision: int) -> None:
        """PRECISION_F32 (default, the parity configuration) or PRECISION_BF16 (trunk + FPN in bf16)."""
        check(lib().ocr_det_set_precision(self._h, precision))

    def synchronize(self) -> None:
        check(lib().ocr_det_synchronize(self._h))

    def forward_host(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, c, h, w = x.shape
        assert c == 1
        prob = np.empty((n, 1, h, w), np.float32)
        check(lib().ocr_det_forward(self._h, _ptr(x), n, h, w, _ptr(prob), MEM_HOST))
        return prob

    def forward_host_u8(self, x: np.ndarray) -> np.ndarray:
        """ocr_det_forward_u8 on a host batch of raw u8 luma frames (the reference's image type)."""
        x = np.ascontiguousarray(x, dtype=np.uint8)
        n, c, h, w = x.shape
        assert c == 1
        prob = np.empty((n, 1, h, w), np.float32)
        check(lib().ocr_det_forward_u8(self._h, _ptr(x), n, h, w, _ptr(prob), MEM_HOST))
        return prob

    def forward_u8_device(self, x_ptr: int, n: int, h: int, w: int, prob_ptr: int) -> None:
        """ocr_det_forward_u8 on device pointers (blocking)."""
        check(lib().ocr_det_forward_u8(self._h, x_ptr, n, h, w, prob_ptr, MEM_DEVICE))

    def detect_pipelined_host(self, x, n: int = 0, h: int = 0, w: int = 0, adjust_values=None, prob_out=None,
                              params: Optional[PostprocParams] = None, convert: bool = True):
        """ocr_det_detect_pipelined_host: x is a host array of frames (float32 or uint8, N x 1 x H x W; a HostBuffer view
        keeps the copy asynchronous) or None to flush.  Returns the PREVIOUS batch's polygons (None on the first call)."""
        adj_p = None
        xp, elem = None, ELEM_F32
        if x is not None:
            assert x.flags["C_CONTIGUOUS"] and x.dtype in (np.float32, np.uint8)
            n, _, h, w = x.shape
            elem = ELEM_U8 if x.dtype == np.uint8 else ELEM_F32
            xp = _ptr(x)
            adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
            adj_p = adj.ctypes.data_as(C.POINTER(C.c_double))
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_detect_pipelined_host(self._h, xp, elem, n, h, w, _ptr(prob_out) if prob_out is not None else None, adj_p,
                                                  C.byref(params) if params is not None else None, C.byref(out)))
        if not out:
            return None
        try:
            return polygons_to_python(out) if convert else (out.contents.n_polygons, out.contents.n_vertices)
        finally:
            lib().ocr_polygons_free(out)

    def forward_device(self, x_ptr: int, n: int, h: int, w: int, prob_ptr: int, bitmap_ptr: int = 0,
                       thresh: float = 0.6) -> None:
        """Enqueue only (device pointers)."""
        check(lib().ocr_det_forward_async(self._h, x_ptr, n, h, w, prob_ptr, bitmap_ptr or None, thresh))

    def forward_profile(self, x_ptr: int, n: int, h: int, w: int, prob_ptr: int):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        cnt = C.c_int(0)
        check(lib().ocr_det_forward_profile(self._h, x_ptr, n, h, w, prob_ptr, cap, names, ms, fl, by, C.byref(cnt)))
        return [(names[i].decode(), float(ms[i]), float(fl[i]), float(by[i])) for i in range(cnt.value)]

    def postprocess_and_crops(self, prob: np.ndarray, frames: np.ndarray, adjust_values: np.ndarray,
                              params: Optional[PostprocParams] = None):
        """get_boxes_and_box_scores followed by the detect -> recognise crop step (host arrays):
        returns (polygons, scores, crops[P x 784])."""
        prob = np.ascontiguousarray(prob, dtype=np.float32)
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        n, _, h, w = prob.shape
        adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
        adj_p = adj.ctypes.data_as(C.POINTER(C.c_double))
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_postprocess(self._h, _ptr(prob), n, h, w, MEM_HOST, adj_p,
                                        C.byref(params) if params is not None else None, C.byref(out)))
        try:
            crops = np.zeros((out.contents.n_polygons, 784), np.float32)
            check(lib().ocr_extract_crops(self._h, _ptr(frames), n, h, w, MEM_HOST, out, adj_p, _ptr(crops)))
            polys, scores = polygons_to_python(out)
            return polys, scores, crops
        finally:
            lib().ocr_polygons_free(out)

    def postprocess_and_crops_device(self, prob_ptr: int, frames_ptr: int, n: int, h: int, w: int, adjust_values,
                                     alloc_crops, params: Optional[PostprocParams] = None):
        """The same on device-resident tensors: prob / frames are device pointers (N x 1 x H x W f32),
        `alloc_crops(P)` returns the device pointer of a P x 784 f32 buffer once the polygon count P is known.
        Returns (polygons, scores, P)."""
        adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
        adj_p = adj.ctypes.data_as(C.POINTER(C.c_double))
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_postprocess(self._h, prob_ptr, n, h, w, MEM_DEVICE, adj_p,
                                        C.byref(params) if params is not None else None, C.byref(out)))
        try:
            npoly = out.contents.n_polygons
            if npoly:
                check(lib().ocr_extract_crops(self._h, frames_ptr, n, h, w, MEM_DEVICE, out, adj_p, alloc_crops(npoly)))
            polys, scores = polygons_to_python(out)
            return polys, scores, npoly
        finally:
            lib().ocr_polygons_free(out)

    def debug_stage(self, stage_id: int, shape_nhwc) -> np.ndarray:
        """Test hook: NHWC intermediate of the last forward, returned as NCHW."""
        n = C.c_size_t(0)
        check(test_lib().ocr_test_det_stage(self._h, stage_id, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        check(test_lib().ocr_test_det_stage(self._h, stage_id, _ptr(out), n.value, C.byref(n)))
        return np.ascontiguousarray(out.reshape(shape_nhwc).transpose(0, 3, 1, 2))

    def preprocess_image(self, rgba: np.ndarray, target_w: int, target_h: int, want_f32: bool = False):
        """image_ops::preprocess_image after decoding: rgba h x w x 4 u8 -> (gray HxW u8[, f32 frame], adj_x, adj_y)."""
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w = rgba.shape[:2]
        gray = np.empty((target_h, target_w), np.uint8)
        f32 = np.empty((1, 1, target_h, target_w), np.float32) if want_f32 else None
        adj = (C.c_double * 2)()
        check(lib().ocr_preprocess_image(self._h, _ptr(rgba), w, h, target_w, target_h, _ptr(gray),
                                         _ptr(f32) if want_f32 else None, adj, MEM_HOST))
        return (gray, f32, adj[0], adj[1]) if want_f32 else (gray, adj[0], adj[1])

    def debug_conv_bench(self, n, h, w, cin, cout, ks=3, stride=1, src_mode=0, iters=5) -> float:
        """Test hook: average milliseconds of one conv_igemm launch of this shape."""
        ms = C.c_float(0.0)
        check(test_lib().ocr_test_conv_bench(self._h, n, h, w, cin, cout, ks, stride, src_mode, iters, C.byref(ms)))
        return ms.value

    def debug_conv_run(self, x_nhwc, wgt_ohwi, stride=1, scale=None, bias=None, residual=None, up_residual=None,
                       relu=False, cat4_shape=None, in_bf16=False, out_bf16=False, want_out=True, want_out2=False, variant=0):
        """One conv_igemm launch on caller data (test hook).  x_nhwc: N x H x W x Cin f32, or with
        cat4_shape=(n, h, w) the flat concatenation p5|p4|p3|p2.  Returns (out, out2) as N x Ho x Wo x Cout f32."""
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        x, wg = f(x_nhwc), f(wgt_ohwi)
        cout, kk, cin = wg.shape
        ks = int(round(kk ** 0.5))
        n, h, w = cat4_shape if cat4_shape else x.shape[:3]
        pad = (ks - 1) // 2
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        out = np.empty((n, ho, wo, cout), np.float32) if want_out else None
        out2 = np.empty((n, ho, wo, cout), np.float32) if want_out2 else None
        sc, bi, rs, ur = f(scale), f(bias), f(residual), f(up_residual)
        p = lambda a: None if a is None else _ptr(a)
        check(test_lib().ocr_test_conv_run(self._h, int(in_bf16), int(out_bf16), _ptr(x), n, h, w, cin, _ptr(wg), cout, ks,
                                           stride, p(sc), p(bi), p(rs), p(ur), int(relu), int(bool(cat4_shape)), int(variant),
                                           p(out), p(out2)))
        return out, out2

    def debug_bf16_basic_block(self, x_nhwc, w1, w2, scale1=None, bias1=None, scale2=None, bias2=None, fused=True, num_cus=0, iters=1):
        """one BasicBlock 64 -> 64 of the bf16 precision (test hook): fused = basic_block_bf16_c64.hip (one launch), otherwise two
        conv3x3_bf16_c64 launches.  x N x H x W x 64 f32 (rounded to bf16 inside), w [64][9][64].  Returns (out f32, ms per block)."""
        x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
        n, h, w, c = x.shape
        assert c == 64
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        p = lambda a: None if a is None else _ptr(a)
        w1, w2, s1, b1, s2, b2 = f(w1), f(w2), f(scale1), f(bias1), f(scale2), f(bias2)
        out = np.empty_like(x)
        ms = C.c_float(0.0)
        check(test_lib().ocr_test_bf16_basic_block(self._h, _ptr(x), n, h, w, _ptr(w1), p(s1), p(b1), _ptr(w2), p(s2), p(b2), int(bool(fused)),
                                                   int(num_cus), int(iters), _ptr(out), C.byref(ms)))
        return out, ms.value

    def debug_gemm_batched(self, x_bmk, w_bnk, variant=2, scale=None, bias=None, relu=False):
        """B independent GEMMs out[b] = x[b] @ w[b].T through conv_igemm's batched 1x1 mode (test hook; what the Winograd GEMMs of layer3 /
        layer4 launch): x B x M x K, w B x N x K -> B x M x N.  variant 0 f32 MFMA, 2 split-bf16 128-wide tiles, 3 the 256 x 128 form."""
        x = np.ascontiguousarray(x_bmk, dtype=np.float32)
        wg = np.ascontiguousarray(w_bnk, dtype=np.float32)
        b, m, k = x.shape
        _, nn, _ = wg.shape
        out = np.empty((b, m, nn), np.float32)
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        sc, bi = f(scale), f(bias)
        p = lambda a: None if a is None else _ptr(a)
        check(test_lib().ocr_test_conv_run(self._h, 0, 0, _ptr(x), 1, 1, m, k, _ptr(wg), nn, 1, 1, p(sc), p(bi), None, None, int(relu), 0,
                                           int(variant) | (b << 8), _ptr(out), None))
        return out

    def debug_winograd_conv(self, x_nhwc, wgt_ohwi, scale=None, bias=None, residual=None, relu=False, unfused=False):
        """3x3 s1 p1 conv through the Winograd path on caller data (test hook): N x H x W x Cout f32."""
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        x, wg = f(x_nhwc), f(wgt_ohwi)
        n, h, w, cin = x.shape
        cout = wg.shape[0]
        assert wg.shape == (cout, 9, cin)
        sc, bi, rs = f(scale), f(bias), f(residual)
        p = lambda a: None if a is None else _ptr(a)
        out = np.empty((n, h, w, cout), np.float32)
        check(test_lib().ocr_test_winograd_conv(self._h, _ptr(x), n, h, w, cin, _ptr(wg), cout, p(sc), p(bi), p(rs),
                                                int(relu), int(unfused), _ptr(out)))
        return out

    def debug_box_scores(self, pred_hw: np.ndarray, polys):
        """Test hook: raw (sum, count) of the GPU box-score kernel for given polygons."""
        pred = np.ascontiguousarray(pred_hw, dtype=np.float32)
        h, w = pred.shape
        xy = np.asarray([c for p in polys for pt in p for c in pt], dtype=np.int32)
        cnt = np.asarray([len(p) for p in polys], dtype=np.int32)
        sums = np.empty(len(polys), np.float64)
        counts = np.empty(len(polys), np.float64)
        check(test_lib().ocr_test_box_scores(self._h, _ptr(pred), h, w, _ptr(xy), _ptr(cnt), len(polys), _ptr(sums),
                                        _ptr(counts)))
        return sums, counts

    def detect_pipelined(self, x_ptr: int, n: int, h: int, w: int, prob_ptr: int, adjust_values=None,
                         params: Optional[PostprocParams] = None, convert: bool = True):
        """ocr_det_detect_pipelined: enqueue this batch's forward, get the PREVIOUS batch's polygons (None on the first
        call); x_ptr = 0 flushes.  convert=False returns (n_polygons, n_vertices) instead of Python lists."""
        adj_p = None
        if x_ptr:
            adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
            adj_p = adj.ctypes.data_as(C.POINTER(C.c_double))
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_detect_pipelined(self._h, x_ptr or None, n, h, w, prob_ptr or None, adj_p,
                                             C.byref(params) if params is not None else None, C.byref(out)))
        if not out:
            return None
        try:
            return polygons_to_python(out) if convert else (out.contents.n_polygons, out.contents.n_vertices)
        finally:
            lib().ocr_polygons_free(out)

    def detect_pipelined_block(self, x_ptr: int, n: int, h: int, w: int, prob_ptr: int, adjust_values=None,
                               params: Optional[PostprocParams] = None):
        """ocr_det_detect_pipelined returning the PREVIOUS batch's ocr_polygons_t block itself (a ctypes pointer, None on the
        first call; release it with free_block) - what extract_crops_block consumes without a detour through Python lists."""
        adj_p = None
        if x_ptr:
            adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
            adj_p = adj.ctypes.data_as(C.POINTER(C.c_double))
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_detect_pipelined(self._h, x_ptr or None, n, h, w, prob_ptr or None, adj_p,
                                             C.byref(params) if params is not None else None, C.byref(out)))
        return out if out else None

    def extract_crops_block(self, block, frames_ptr: int, n: int, h: int, w: int, adjust_values, crops_ptr: int) -> int:
        """ocr_extract_crops for a polygon block on device-resident frames; returns the number of crops written."""
        adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
        if block.contents.n_polygons:
            check(lib().ocr_extract_crops(self._h, frames_ptr, n, h, w, MEM_DEVICE, block, adj.ctypes.data_as(C.POINTER(C.c_double)), crops_ptr))
        return block.contents.n_polygons

    @staticmethod
    def free_block(block) -> None:
        if block:
            lib().ocr_polygons_free(block)

    def postprocess_counts(self, prob, n: int, h: int, w: int, adjust_values: np.ndarray, mem_kind: int = MEM_HOST,
                           params: Optional[PostprocParams] = None) -> Tuple[int, int]:
        """get_boxes_and_box_scores without turning the CSR block into Python objects: (polygons, vertices).
        What a benchmark times when it wants the library, not the harness."""
        adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_postprocess(self._h, _ptr(prob), n, h, w, mem_kind,
                                        adj.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(params) if params is not None else None, C.byref(out)))
        try:
            return out.contents.n_polygons, out.contents.n_vertices
        finally:
            lib().ocr_polygons_free(out)

    def post_stats(self) -> dict:
        """Cumulative counters of where this handle's polygon chain ran (ocr_det_post_stats)."""
        a = (C.c_int64 * 6)()
        check(lib().ocr_det_post_stats(self._h, a))
        return dict(zip(("images_device_traced", "images_host_traced", "candidates_device", "candidates_host", "images_device_chain", "passes"),
                        (int(v) for v in a)))

    def postprocess(self, prob, n: int, h: int, w: int, adjust_values: np.ndarray, mem_kind: int = MEM_HOST,
                    params: Optional[PostprocParams] = None):
        adj = np.ascontiguousarray(adjust_values, dtype=np.float64).reshape(n, 2)
        out = C.POINTER(Polygons)()
        check(lib().ocr_det_postprocess(self._h, _ptr(prob), n, h, w, mem_kind,
                                        adj.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(params) if params is not None else None, C.byref(out)))
        try:
            return polygons_to_python(out)
        finally:
            lib().ocr_polygons_free(out)


class Recognizer:
    """Owns an ocr_rec_t.  Mirrors `Net::new(&weights.root())` + `weights.load(..)`
    (/root/reference/src/char_recognition/mod.rs:44-46)."""

    def __init__(self, weights_blob: Optional[bytes], device: int = 0, varstore_path: Optional[str] = None,
                 options: Optional[str] = None):
        self._h = C.c_void_p()
        if varstore_path is not None:
            check(lib().ocr_rec_create_from_varstore(os.fsencode(varstore_path), device, C.byref(self._h)))
        else:
            check(lib().ocr_rec_create(weights_blob, len(weights_blob), device, C.byref(self._h)))
        if options:
            self.set_options(options)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            lib().ocr_rec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, raw_stream: Optional[int]) -> None:
        check(lib().ocr_rec_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    def synchronize(self) -> None:
        check(lib().ocr_rec_synchronize(self._h))

    def set_options(self, options: str) -> None:
        """`small_batch=0`: every batch on the throughput kernels (bit-exact batch-size invariance)."""
        check(lib().ocr_rec_set_options(self._h, options.encode()))

    def forward_host(self, crops: np.ndarray) -> np.ndarray:
        crops = np.ascontiguousarray(crops, dtype=np.float32).reshape(-1, 784)
        n = crops.shape[0]
        logits = np.empty((n, 62), np.float32)
        check(lib().ocr_rec_forward(self._h, _ptr(crops), n, _ptr(logits), MEM_HOST))
        return logits

    def classify_host(self, crops: np.ndarray):
        crops = np.ascontiguousarray(crops, dtype=np.float32).reshape(-1, 784)
        n = crops.shape[0]
        labels = np.empty(n, np.int32)
        probs = np.empty(n, np.float64)
        check(lib().ocr_rec_classify(self._h, _ptr(crops), n, _ptr(labels), _ptr(probs), MEM_HOST))
        return labels, probs

    def ctc_greedy_decode(self, logits: np.ndarray, blank: int):
        """EXTENSION (no reference counterpart): logits N x T x C f32 in host memory -> (labels N x T int32 padded with -1, lengths N)."""
        x = np.ascontiguousarray(logits, dtype=np.float32)
        n, t, c = x.shape
        labels = np.empty((n, t), np.int32)
        lengths = np.empty(n, np.int32)
        check(lib().ocr_ctc_greedy_decode(self._h, _ptr(x), n, t, c, int(blank), MEM_HOST, _ptr(labels), _ptr(lengths)))
        return labels, lengths

    def ctc_greedy_decode_device(self, logits_ptr: int, n: int, t: int, c: int, blank: int, labels_ptr: int, lengths_ptr: int) -> None:
        check(lib().ocr_ctc_greedy_decode(self._h, C.c_void_p(logits_ptr), n, t, c, int(blank), MEM_DEVICE, C.c_void_p(labels_ptr), C.c_void_p(lengths_ptr)))

    def classify_profile(self, crops_ptr: int, n: int, labels_ptr: int = 0, probs_ptr: int = 0):
        """[(kernel, ms, executed flops, bytes)] of one classify pass (device pointers)."""
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        cnt = C.c_int(0)
        check(lib().ocr_rec_classify_profile(self._h, crops_ptr, n, labels_ptr or None, probs_ptr or None, cap, names, ms,
                                             fl, by, C.byref(cnt)))
        return [(names[i].decode(), float(ms[i]), float(fl[i]), float(by[i])) for i in range(cnt.value)]

    def classify_device(self, crops_ptr: int, n: int, logits_ptr: int, labels_ptr: int, probs_ptr: int) -> None:
        check(lib().ocr_rec_classify_async(self._h, crops_ptr, n, logits_ptr or None, labels_ptr or None,
                                           probs_ptr or None))


def python_to_polygons(polys, scores):
    """PolygonScores as Python lists -> (Polygons struct, keep-alive arrays) for the calls that take a block."""
    img = np.cumsum([0] + [len(p) for p in polys]).astype(np.int32)
    flat = [pg for p in polys for pg in p]
    po = np.cumsum([0] + [len(pg) for pg in flat]).astype(np.int32)
    xy = np.asarray([c for pg in flat for v in pg for c in v], dtype=np.uint32)
    sc = np.asarray([s for ss in scores for s in ss], dtype=np.float64)
    st = Polygons(len(polys), len(flat), int(po[-1]), img.ctypes.data_as(C.POINTER(C.c_int32)),
                  po.ctypes.data_as(C.POINTER(C.c_int32)), xy.ctypes.data_as(C.POINTER(C.c_uint32)),
                  sc.ctypes.data_as(C.POINTER(C.c_double)))
    return st, (img, po, xy, sc)


class Comm:
    """Owns an ocr_comm_t: the RCCL communicator behind the C ABI (one per rank; creation is collective)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(lib().ocr_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def rccl_version() -> int:
        v = C.c_int(0)
        check(lib().ocr_comm_rccl_version(C.byref(v)))
        return v.value

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = 0):
        self._h = C.c_void_p()
        self.world, self.rank = world, rank
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().ocr_comm_create(buf, world, rank, device, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            lib().ocr_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather_polygons(self, polys, scores):
        st, keep = python_to_polygons(polys, scores)
        out = C.POINTER(Polygons)()
        check(lib().ocr_comm_all_gather_polygons(self._h, C.byref(st), C.byref(out)))
        try:
            return polygons_to_python(out)
        finally:
            lib().ocr_polygons_free(out)

    def all_gather_labels(self, labels: np.ndarray, capacity: int):
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        out = np.empty(capacity, np.int32)
        counts = np.zeros(self.world, np.int32)
        n = C.c_int(0)
        check(lib().ocr_comm_all_gather_labels(self._h, _ptr(labels), labels.size, _ptr(out), capacity, _ptr(counts), C.byref(n)))
        return out[:n.value].copy(), counts


def comm_assemble(shards):
    """Test hook: what ocr_comm_all_gather_polygons returns for these per-rank (polys, scores) shards, without RCCL."""
    structs, keep = [], []
    for polys, scores in shards:
        st, k = python_to_polygons(polys, scores)
        structs.append(st)
        keep.append(k)
    arr = (C.POINTER(Polygons) * len(structs))(*[C.pointer(s) for s in structs])
    out = C.POINTER(Polygons)()
    check(test_lib().ocr_test_comm_assemble(arr, len(structs), C.byref(out)))
    try:
        return polygons_to_python(out)
    finally:
        lib().ocr_polygons_free(out)


# ---- host-geometry hooks (CPU only; used by tests to pin the C++ geometry to the KATs)
def host_contour_candidates(bitmap01: np.ndarray) -> List[List[Tuple[int, int]]]:
    bm = np.ascontiguousarray(bitmap01, dtype=np.uint8)
    h, w = bm.shape
    max_pts, max_polys = 1 << 20, 1 << 16
    xy = np.empty(2 * max_pts, np.int32)
    cnt = np.empty(max_polys, np.int32)
    n = C.c_int(0)
    check(test_lib().ocr_test_contour_candidates(_ptr(bm), h, w, _ptr(xy), _ptr(cnt), max_pts, max_polys, C.byref(n)))
    out, pos = [], 0
    for k in range(n.value):
        c = int(cnt[k])
        out.append([(int(xy[2 * (pos + i)]), int(xy[2 * (pos + i) + 1])) for i in range(c)])
        pos += c
    return out


def _contours_call(fn, bitmap01, extra_status=False, max_pts=1 << 20, max_polys=1 << 16, tail=()):
    bm = np.ascontiguousarray(bitmap01, dtype=np.uint8)
    h, w = bm.shape
    xy = np.empty(2 * max_pts, np.int32)
    cnt = np.empty(max_polys, np.int32)
    n, st = C.c_int(0), C.c_int(0)
    args = [_ptr(bm), h, w, _ptr(xy), _ptr(cnt), max_pts, max_polys, C.byref(n)] + ([C.byref(st)] if extra_status else []) + list(tail)
    check(fn(*args))
    out, pos = [], 0
    for k in range(n.value):
        c = int(cnt[k])
        out.append([(int(xy[2 * (pos + i)]), int(xy[2 * (pos + i) + 1])) for i in range(c)])
        pos += c
    return (out, st.value) if extra_status else out


def host_contours(bitmap01: np.ndarray):
    """Raw contours of the host tracer (postproc_geom.cpp::find_contours)."""
    L = test_lib()
    L.ocr_test_host_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return _contours_call(L.ocr_test_host_contours, bitmap01)


def device_contours(bitmap01: np.ndarray, max_pts: int = 1 << 20, max_polys: int = 1 << 16, sequential: bool = False):
    """Raw contours of the device tracer (contours.hip; the parallel form, or the one-wave-per-image form) and its status
    (0 ok, 1 buffers too small, 2 guard, 3 parallel form: a start outside its list of plausible starts)."""
    L = test_lib()
    L.ocr_test_device_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    return _contours_call(L.ocr_test_device_contours, bitmap01, True, max_pts, max_polys, tail=(int(sequential),))


def device_candidates(contours, h: int, w: int, max_pts: int = 1 << 18, max_polys: int = 1 << 14):
    """candidates.hip on the given contours of one h x w map (lists of (x, y)): the candidate polygons after arc length, Douglas-Peucker
    and the >= 4 points filter, in list order, and their clamped job boxes (min_x, min_y, bw, bh)."""
    L = test_lib()
    L.ocr_test_device_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.POINTER(C.c_int)]
    lens = np.asarray([len(c) for c in contours] or [0], np.int32)
    flat = np.asarray([v for c in contours for q in c for v in q] or [0, 0], np.int32)
    xy = np.empty(2 * max_pts, np.int32)
    cnt = np.empty(max_polys, np.int32)
    box = np.empty(4 * max_polys, np.int32)
    n = C.c_int(0)
    check(L.ocr_test_device_candidates(_ptr(flat), _ptr(lens), len(contours), h, w, _ptr(xy), _ptr(cnt), _ptr(box), max_pts, max_polys, C.byref(n)))
    out, boxes, pos = [], [], 0
    for k in range(n.value):
        c = int(cnt[k])
        out.append([(int(xy[2 * (pos + i)]), int(xy[2 * (pos + i) + 1])) for i in range(c)])
        boxes.append(tuple(int(v) for v in box[4 * k:4 * k + 4]))
        pos += c
    return out, boxes


def host_expand_polygon(pts: Sequence[Tuple[int, int]], factor: float = 2.0):
    a = np.asarray(pts, dtype=np.int32).reshape(-1)
    out = np.empty(8192, np.int32)
    n = C.c_int(0)
    ss = C.c_double(0.0)
    check(test_lib().ocr_test_expand_polygon(_ptr(a), len(pts), factor, _ptr(out), 4096, C.byref(n), C.byref(ss)))
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)], ss.value


def host_min_area_box(pts: Sequence[Tuple[int, int]]):
    a = np.asarray(pts, dtype=np.int32).reshape(-1)
    box = np.empty(8, np.int32)
    ss = C.c_double(0.0)
    check(test_lib().ocr_test_min_area_box(_ptr(a), len(pts), _ptr(box), C.byref(ss)))
    return [(int(box[2 * i]), int(box[2 * i + 1])) for i in range(4)], ss.value

"""Synthetic weights and the flat weight blob the C-ABI consumes.

The reference ships no trained weights (/root/reference/.gitignore:16), so parity and
benchmarks run on seeded synthetic parameters.  Names and shapes follow the
VarStore layout of /root/reference/src/text_detection/model.rs:65-105 and
/root/reference/src/char_recognition/model.rs:13-24.

Blob layout (little endian):
    0   char[4]  "OCRW"
    4   u32      version (1)
    8   u32      n_tensors
    12  u32      reserved
    16  n_tensors x record{ char name[64]; u32 ndim; u32 dims[4]; u64 offset; u64 count }
    ..  f32 data, each tensor 64-byte aligned, `offset` is from the blob start
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

_REC = struct.Struct("<64sI4IQQ")


def det_param_specs() -> List[Tuple[str, Tuple[int, ...]]]:
    specs: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix: str, c: int) -> None:
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            specs.append((f"{prefix}.{leaf}", (c,)))

    specs.append(("conv1.weight", (64, 1, 7, 7)))
    bn("bn1", 64)
    cin = 64
    for li, cout in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"layer{li}.{b}"
            c_in = cin if b == 0 else cout
            specs.append((f"{p}.conv1.weight", (cout, c_in, 3, 3)))
            bn(f"{p}.bn1", cout)
            specs.append((f"{p}.conv2.weight", (cout, cout, 3, 3)))
            bn(f"{p}.bn2", cout)
            if b == 0 and li > 1:
                specs.append((f"{p}.downsample.0.weight", (cout, c_in, 1, 1)))
                bn(f"{p}.downsample.1", cout)
        cin = cout
    specs += [("in5.weight", (256, 512, 1, 1)), ("in4.weight", (256, 256, 1, 1)),
              ("in3.weight", (256, 128, 1, 1)), ("in2.weight", (256, 64, 1, 1))]
    for n in ("out5", "out4", "out3", "out2", "bin_conv1"):
        specs.append((f"{n}.weight", (64, 256, 3, 3)))
    bn("bin_bn1", 64)
    specs += [("bin_conv_tr1.weight", (64, 64, 2, 2)), ("bin_conv_tr1.bias", (64,))]
    bn("bin_bn2", 64)
    specs += [("bin_conv_tr2.weight", (64, 1, 2, 2)), ("bin_conv_tr2.bias", (1,))]
    return specs


def rec_param_specs() -> List[Tuple[str, Tuple[int, ...]]]:
    return [("conv1.weight", (32, 1, 5, 5)), ("conv1.bias", (32,)),
            ("conv2.weight", (64, 32, 5, 5)), ("conv2.bias", (64,)),
            ("fc1.weight", (512, 1024)), ("fc1.bias", (512,)),
            ("fc2.weight", (62, 512)), ("fc2.bias", (62,))]


def _uniform01(seed: int, tensor_index: int, count: int) -> np.ndarray:
    """Counter-based generator (splitmix64 finaliser): 24-bit uniforms in [0,1)."""
    with np.errstate(over="ignore"):
        i = np.arange(count, dtype=np.uint64)
        z = i + np.uint64((seed * 0x9E3779B97F4A7C15 + (tensor_index + 1) * 0xD1B54A32D192ED03) % (1 << 64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))).astype(np.float32)


def synth_image_batch(seed: int, n: int, h: int, w: int) -> np.ndarray:
    """Synthetic detection frames: integers 0..255 as f32 (the reference feeds raw
    0..255 luma, /root/reference/src/text_detection/mod.rs:46-54), N x 1 x H x W."""
    u = _uniform01(seed, 1000, n * h * w)
    return np.floor(u * 256.0).astype(np.float32).reshape(n, 1, h, w)


def synth_crops(seed: int, n: int) -> np.ndarray:
    """Synthetic recognition crops: U{0..255}/255 as f32, N x 784
    (/root/reference/src/image_ops.rs:73-85 divides by 255)."""
    u = _uniform01(seed, 2000, n * 784)
    keep = _uniform01(seed, 2001, n * 784) < np.float32(0.06)   # sparse "strokes": varied labels
    return (np.where(keep, np.floor(u * 256.0), 0.0).astype(np.float32) / np.float32(255.0)).reshape(n, 784)


# Gains that keep synthetic activations O(1) from raw 0..255 input to the logits
# (measured once with oracle/torch_ref.py; they are part of the synthetic
# distribution, not of the reference).
_DET_GAIN = {"conv1": 1.0 / 128.0, "out5": 0.25, "out4": 0.25, "out3": 0.25, "out2": 0.25,
             "bin_conv_tr2": 0.5}


def make_det_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Kaiming-uniform convolutions, BN gamma ~ U(0.5,1.5), beta ~ U(-.1,.1),
    running_mean ~ U(-.1,.1), running_var ~ U(0.5,1.5): every BN term is
    exercised (tch's own defaults, mean 0 / var 1 / beta 0, would hide bugs).
    The last transposed convolution is scaled so that sigmoid outputs straddle
    the 0.6 binarisation threshold on synthetic frames."""
    out: Dict[str, np.ndarray] = {}
    for ti, (name, shape) in enumerate(det_param_specs()):
        count = int(np.prod(shape))
        u = _uniform01(seed, ti, count)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "weight" and len(shape) == 4:
            if name.startswith("bin_conv_tr"):
                fan_in = shape[0]  # [Cin, Cout, kh, kw], one tap per output pixel
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            bound = np.sqrt(6.0 / fan_in)
            if name.startswith("in") or name.startswith("out"):
                bound = np.sqrt(3.0 / fan_in)  # linear layers (no ReLU after them)
            v = (u * 2.0 - 1.0) * np.float32(bound) * np.float32(_DET_GAIN.get(name.split(".")[0], 1.0))
        elif leaf == "weight":                    # BN gamma
            v = u + np.float32(0.5)
            if name.startswith("layer") and ".bn2." in name:
                v = v * np.float32(0.4)           # damp the residual branch
        elif leaf == "bias" and name.startswith("bin_conv_tr"):
            v = (u * 2.0 - 1.0) * np.float32(0.1)
        elif leaf in ("bias", "running_mean"):
            v = (u * 2.0 - 1.0) * np.float32(0.1)
        elif leaf == "running_var":
            v = u + np.float32(0.5)
        else:
            raise AssertionError(name)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def make_det_weights_text(gain: float = 0.06, tau: float = 120.0, noise: float = 0.02, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic detector weights whose probability map FOLLOWS THE INK of the page: a hand-built signal path
    through the reference graph (model.rs:107-151) plus the random weights of make_det_weights scaled by `noise`,
    so that every channel carries data but bright word boxes on a dark page come out as text-like blobs.  No
    trained weights exist (.gitignore:16); random ones give noise maps, on which polygon lists of two precisions
    cannot be compared - this set is what the end-to-end tests of BASELINE configs[4] run on.

    Signal path (channel 0 everywhere): conv1 = 7x7 box mean -> bn1 identity -> ReLU -> maxpool; the residual
    blocks pass it through (block convs ~ 0, shortcut identity or the 1x1 s2 downsample picking channel 0);
    in2 / out2 / bin_conv1 pick it (centre taps), both transposed convs replicate it, the last one maps it to
    sigmoid(gain * (v - tau)).  Batch norms are identities (gamma 1, beta 0, mean 0, var 1)."""
    rnd = make_det_weights(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in det_param_specs():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "running_var" or (leaf == "weight" and len(shape) == 1):
            out[name] = np.ones(shape, np.float32)
        elif len(shape) == 1:
            out[name] = np.zeros(shape, np.float32)
        else:
            out[name] = (rnd[name] * np.float32(noise)).astype(np.float32)
    out["conv1.weight"][0, 0] = np.float32(1.0 / 49.0)
    for li in (2, 3, 4):
        out[f"layer{li}.0.downsample.0.weight"][0, 0, 0, 0] = 1.0
    out["in2.weight"][0, 0, 0, 0] = 1.0
    out["out2.weight"][0, 0, 1, 1] = 1.0
    out["bin_conv1.weight"][0, 192, 1, 1] = 1.0          # channel 0 of p2 = channel 192 of cat[p5, p4, p3, p2]
    out["bin_conv_tr1.weight"][0, 0] = 1.0
    out["bin_conv_tr2.weight"][:] = 0.0
    out["bin_conv_tr2.weight"][0, 0] = np.float32(gain)
    out["bin_conv_tr2.bias"][0] = np.float32(-gain * tau)
    return out


def synth_text_pages(seed: int, n: int, h: int, w: int, dense: bool = False):
    """Synthetic pages for the end-to-end path: dark noisy paper (10..50) with a jittered grid of slanted word
    boxes (190..250), as f32 N x 1 x H x W.  Returns (frames, boxes) with boxes[i] = [(x0, y0, x1, y1), ...].
    dense: smaller words on a tighter grid (about 65 per 640 x 640 page instead of 18) - the post-processing stress case."""
    rng = np.random.RandomState(seed)
    frames = np.empty((n, 1, h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    boxes = []
    sy, sx, my, mx = (48, 100, 44, 96) if dense else (80, 168, 64, 150)
    for i in range(n):
        page = rng.randint(10, 51, (h, w)).astype(np.float32)
        bl = []
        for gy in range(24, h - my, sy):
            for gx in range(24, w - mx, sx):
                if rng.rand() < 0.15:
                    continue
                if dense:
                    bw, bh = 44 + rng.randint(0, 28), 16 + rng.randint(0, 10)
                    x0, y0 = gx + rng.randint(0, 10), gy + rng.randint(0, 6)
                else:
                    bw, bh = 72 + rng.randint(0, 48), 24 + rng.randint(0, 16)
                    x0, y0 = gx + rng.randint(0, 16), gy + rng.randint(0, 12)
                sl = rng.uniform(-0.08, 0.08)
                m = (xx >= x0) & (xx < x0 + bw) & (yy >= y0 + sl * (xx - x0)) & (yy < y0 + bh + sl * (xx - x0))
                page[m] = rng.randint(190, 251, int(m.sum())).astype(np.float32)
                bl.append((x0, y0, x0 + bw, y0 + bh))
        frames[i, 0] = page
        boxes.append(bl)
    return frames, boxes


def make_rec_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for ti, (name, shape) in enumerate(rec_param_specs()):
        count = int(np.prod(shape))
        u = _uniform01(seed, 500 + ti, count)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else None
        if fan_in is not None:
            gain = 8.0 if name == "fc2.weight" else 1.4   # spreads the 62 logits (top-2 gap >> fp32 noise)
            v = (u * 2.0 - 1.0) * np.float32(np.sqrt(3.0 / fan_in) * gain)
        else:
            v = (u * 2.0 - 1.0) * np.float32(0.01)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def pack_blob(params: Dict[str, np.ndarray]) -> bytes:
    names = list(params.keys())
    header = 16 + _REC.size * len(names)
    off = (header + 63) // 64 * 64
    recs, chunks = [], []
    for name in names:
        a = np.ascontiguousarray(params[name], dtype=np.float32)
        dims = list(a.shape) + [1] * (4 - a.ndim)
        recs.append(_REC.pack(name.encode(), a.ndim, *dims, off, a.size))
        chunks.append((off, a.tobytes()))
        off = (off + a.size * 4 + 63) // 64 * 64
    buf = bytearray(off)
    buf[0:16] = struct.pack("<4sIII", b"OCRW", 1, len(names), 0)
    pos = 16
    for r in recs:
        buf[pos:pos + _REC.size] = r
        pos += _REC.size
    for o, data in chunks:
        buf[o:o + len(data)] = data
    return bytes(buf)


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    magic, ver, n, _ = struct.unpack_from("<4sIII", blob, 0)
    if magic != b"OCRW" or ver != 1:
        raise ValueError("not an OCRW v1 weight blob")
    out: Dict[str, np.ndarray] = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off, count = _REC.unpack_from(blob, 16 + i * _REC.size)
        shape = (d0, d1, d2, d3)[:ndim]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=np.float32, count=count, offset=off).reshape(shape)
    return out


def rename_tch_rec(named: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The recogniser's real VarStore names.  `Net::new` creates conv1, conv2, fc1 and fc2 on the SAME
    `nn::Path` (/root/reference/src/char_recognition/model.rs:13-24), so every layer asks for the names
    "weight" and "bias"; tch 0.3.0's `Path::add` de-duplicates a taken name as `{name}__{n}` with n = the
    number of variables registered so far: weight, bias, weight__2, bias__3, ..., bias__7 (or with bias
    first inside a layer - the order tch's conv / linear constructors register them in).  The eight
    shapes are all different, so each tensor is identified by its shape, whatever the order was.
    A dict that already uses conv1.weight ... fc2.bias (this library's own blobs) passes through."""
    specs = rec_param_specs()
    if all(n in named for n, _ in specs):
        return named
    by_shape = {tuple(shape): name for name, shape in specs}
    out: Dict[str, np.ndarray] = {}
    for name, t in named.items():
        base = name.split("__")[0]
        want = by_shape.get(tuple(t.shape))
        if base in ("weight", "bias") and want is not None and want.endswith("." + base) and want not in out:
            out[want] = t
        else:
            out[name] = t
    return out


def load_varstore(path: str, kind: str | None = None) -> Dict[str, np.ndarray]:
    """Named tensors of a tch `VarStore::save` file (what `vs.load(file)` reads back,
    /root/reference/src/text_detection/mod.rs:41-44, char_recognition/mod.rs:46, utils.rs:55-63).

    tch 0.3.0 writes such a file through torch-sys' `at_save_multi`: one
    `torch::serialize::OutputArchive`, every variable `write(name, tensor)`-n under its dotted
    VarStore path, `save_to(file)` - i.e. a TorchScript-module zip whose parameters carry the
    VarStore names.  `torch.jit.load` opens exactly that container (libtorch keeps reading archives of
    older versions), so the import is: load, walk parameters and buffers, convert to f32 numpy.
    `kind` = "det" / "rec" additionally checks names and shapes against the graph this library
    runs (det_param_specs / rec_param_specs) and fails with the list of what is missing or misshapen.

    Not pinned by the reference: it ships no weight file (.gitignore:16).  Pinned here against
    an archive written by the same libtorch C++ calls (tests/golden/make_varstore_fixture.cpp)."""
    import torch  # plumbing only: the archive reader

    module = torch.jit.load(path, map_location="cpu")
    out: Dict[str, np.ndarray] = {}
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        out[name] = np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy())
    if kind == "rec":
        out = rename_tch_rec(out)
    if kind is not None:
        specs = {"det": det_param_specs, "rec": rec_param_specs}[kind]()
        problems = []
        for name, shape in specs:
            if name not in out:
                problems.append(f"missing {name} {shape}")
            elif tuple(out[name].shape) != tuple(shape):
                problems.append(f"{name}: shape {tuple(out[name].shape)}, expected {shape}")
        if problems:
            raise ValueError(f"{path} is not a {kind} VarStore of this graph: " + "; ".join(problems[:8])
                             + (f" (+{len(problems) - 8} more)" if len(problems) > 8 else ""))
    return out

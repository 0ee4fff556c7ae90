#!/usr/bin/env python3
"""Benchmark of the detection hot path on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the detector (stem .. sigmoid, with the fused binarize) over
one batch of 32 synthetic 640x640 frames that are already resident in HBM, per GPU
(weak scaling: every rank owns its own batch and its own replica of the 48.7 MB
weights; there is no data-path collective - results are gathered once, after the
timed region, to show the exchange step).  Prints ONE JSON line on rank 0.

Extra objects in that line:
  roofline      dominant kernel (largest summed time), measured live with HIP events
                on the launch stream: algorithmic FLOPs of its launches / their time,
                against the 157.3 TF/s dense f32 MFMA peak of MI355X.
  cpu_baseline  the same graph on the host cores through ATen-CPU (oracle/torch_ref.py,
                the operator library the reference reaches through tch), bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ocr_rs_amd  # noqa: E402,F401
from ocr_rs_amd import capi  # noqa: E402
from ocr_rs_amd import weights as W  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (the headline vendor figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0
GFLOP_PER_640_IMAGE = 48.365568  # SURVEY.md section 8(d) / BASELINE.md section 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip recognition / post-processing side numbers")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the reference's arithmetic (BASELINE configs[1], the headline); bf16 = the opt-in "
                         "OCR_PRECISION_BF16 trunk/FPN (configs[4]), reported as its own line")
    return ap.parse_args()


def pmc_traffic(kernel: str, n: int, s: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md);
    None when no extract for this workload is committed.  Not measured live: PMC needs rocprofv3."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        if t.get("batch") == n and t.get("size") == s:
            e = t["kernels"].get(kernel)
            return (None, None) if e is None else (e["bytes_per_launch"], t["source"])
    except Exception:
        pass
    return None, None


def text_like_maps(n: int, s: int, seed: int) -> np.ndarray:
    """Probability maps with text-like blobs (the reference's gt_shrinked fixtures, cropped to s x s and
    jittered): random-weight network outputs are noise, which is not what post-processing sees in use."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    names = ["gt_shrinked_img55.png", "gt_shrinked_img224.png", "gt_shrinked_img494.png", "gt_shrinked_img545.png"]
    out = []
    for i in range(n):
        g = np.array(Image.open(os.path.join(ROOT, "tests", "golden", names[i % 4])).convert("L"))
        o = (800 - s) // 2
        g = g[o:o + s, o:o + s] if s <= 800 else np.pad(g, ((0, s - 800), (0, s - 800)))
        out.append(np.where(g > 127, 0.8 + 0.2 * rng.rand(s, s), 0.1 * rng.rand(s, s)).astype(np.float32))
    return np.ascontiguousarray(np.stack(out)[:, None])


def host_cores() -> int:
    """CPU share of this container: cgroup quota if any (a 1-GPU box gets 16), else affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16) if n > 64 else n


def cpu_baseline(det_w, size: int, budget_s: float):
    """Reference stand-in on the host cores: same graph through ATen CPU kernels."""
    from oracle import torch_ref as T
    cores = host_cores()
    torch.set_num_threads(cores)
    x = W.synth_image_batch(1, 2, size, size)
    T.det_forward(det_w, x[:1])  # warm the thread pool / allocator
    done, t0 = 0, time.perf_counter()
    while True:
        T.det_forward(det_w, x)
        done += x.shape[0]
        el = time.perf_counter() - t0
        if el > budget_s or done >= 64:
            break
    return {"value": round(done / el, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{done} frames of {size}x{size} f32 in batches of 2 through oracle/torch_ref.py "
                      f"(ATen CPU, {cores} threads), {el:.1f} s"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU; OCR_BENCH_BACKEND=gloo (+ fewer GPUs than ranks) only exists to rehearse the
    # multi-rank control flow on a 1-GPU box
    backend = os.environ.get("OCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    n, s = a.batch, a.size
    det_w = W.make_det_weights(0)
    det = capi.Detector(W.pack_blob(det_w), local)
    if a.dtype == "bf16":
        det.set_precision(capi.PRECISION_BF16)
    peak = BF16_MFMA_PEAK_TFLOPS if a.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
    stream = torch.cuda.Stream(device=local)
    det.set_stream(stream.cuda_stream)
    x = torch.from_numpy(W.synth_image_batch(1 + rank, n, s, s)).to(f"cuda:{local}")
    prob = torch.empty_like(x)
    bitmap = torch.empty(x.shape, dtype=torch.uint8, device=x.device)

    def step():
        det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr(), bitmap.data_ptr(), 0.6)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(a.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=x.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel: HIP events around every launch, on the launch stream
    roof = None
    executed_gflop = None
    if rank == 0:
        agg = {}
        reps = 3
        for _ in range(reps):
            for name, ms, fl, by in det.forward_profile(x.data_ptr(), n, s, s, prob.data_ptr()):
                e = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
                e[0] += ms
                e[1] += fl
                e[2] += by
                e[3] += 1
        dom = max(agg.items(), key=lambda kv: kv[1][0])
        name, (ms, fl, by, cnt) = dom
        executed = fl / (ms * 1e-3) / 1e12
        # the engine reports the FLOPs a kernel executes.  A fused Winograd F(2x2,3x3) launch is a 3x3 conv whose
        # algorithmic work (2 M N 9C, what SURVEY 8d counts) is 36/16 of the multiplies it executes; `achieved`
        # is algorithmic work / time as the contract defines it (so it can exceed the MFMA peak of the direct
        # algorithm), `executed` is what the matrix cores actually do.
        alg_factor = 36.0 / 16.0 if name.startswith("winograd_fused") else 1.0
        achieved = executed * alg_factor
        executed_gflop = sum(v[1] for v in agg.values()) / reps / 1e9
        traffic, traffic_source = pmc_traffic(name, n, s)
        roof = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "executed": {"tflops": round(executed, 2), "frac": round(executed / peak, 4),
                             "note": "MFMA FLOPs the kernel executes / time" + (
                                 "; Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs instead of 36" if alg_factor > 1 else "")},
                "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_source,
                "launches_per_step": cnt // reps, "avg_launch_ms": round(ms / cnt, 4),
                "avg_launch_gflop": round(fl * alg_factor / cnt / 1e9, 3),
                "all_kernels": {k: {"ms_per_step": round(v[0] / reps, 4),
                                    "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else None,
                                    "gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None}
                                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}

    # ---- the exchange step of the sharded path (outside the timed region): every rank post-processes
    # text-like maps of its shard (get_boxes_and_box_scores) and the variable-length polygon blocks
    # are all-gathered - RCCL over xGMI when the backend is nccl (ocr-rs_amd/parallel.py)
    gathered = None
    post = {}
    try:
        maps = text_like_maps(min(n, 8), s, seed=rank)
        pm = torch.from_numpy(maps).to(x.device)
        torch.cuda.synchronize()
        params = capi.default_params(skip_degenerate=True)
        adj = np.ones((maps.shape[0], 2))
        polys, scores = det.postprocess(pm, maps.shape[0], s, s, adj, capi.MEM_DEVICE, params)
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            det.postprocess(pm, maps.shape[0], s, s, adj, capi.MEM_DEVICE, params)
        post = {"postprocess_images_per_s": round(maps.shape[0] * reps / (time.perf_counter() - t1), 1),
                "postprocess_polygons_per_image": round(sum(len(p) for p in polys) / maps.shape[0], 2)}
        if dist is not None:
            from ocr_rs_amd import parallel as P
            all_p, all_s = P.all_gather_results(polys, scores, x.device if backend == "nccl" else torch.device("cpu"))
            gathered = {"images": len(all_p), "polygons": sum(len(p) for p in all_p)}
    except Exception as e:  # side numbers never hide the headline
        post = {"postprocess_error": str(e)}

    extras = {}
    if rank == 0 and not a.no_extras:
        try:
            rec_w = W.make_rec_weights(0)
            rec = capi.Recognizer(W.pack_blob(rec_w), local)
            rec.set_stream(stream.cuda_stream)
            nc = 256
            crops = torch.from_numpy(W.synth_crops(2, nc)).to(x.device)
            labels = torch.empty(nc, dtype=torch.int32, device=x.device)
            probs = torch.empty(nc, dtype=torch.float64, device=x.device)
            with torch.cuda.stream(stream):
                for _ in range(3):
                    rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                it = 50
                for _ in range(it):
                    rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
                torch.cuda.synchronize()
                extras["rec_crops_per_s_b256"] = round(nc * it / (time.perf_counter() - t1), 1)
        except Exception as e:  # side numbers never hide the headline
            extras["rec_error"] = str(e)

    total_images = n * world * a.steps
    if rank == 0:
        line = {
            "metric": "images/sec (640x640 detect)", "value": round(total_images / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"detection forward (ResNet18+FPN+prob head, fused binarize), batch {n} x 1x{s}x{s} "
                                   f"f32 frames per GPU, BASELINE configs[1]"
                                   + (" in the opt-in bf16 precision of configs[4]" if a.dtype == "bf16" else ""),
                       "global_batch": n * world, "frame": [s, s], "parallelism": f"replica x{world}, frames sharded"},
            # the reference graph's layer-by-layer work per second, and what the kernels execute
            # after folding the FPN laterals / upsampled concat quarters into phase convs
            "tflops_reference_graph": round(total_images * GFLOP_PER_640_IMAGE * (s * s) / (640 * 640) / elapsed / 1e3, 2),
            "tflops_executed": None if executed_gflop is None else round(executed_gflop * a.steps * world / elapsed / 1e3, 2),
            "roofline": roof,
        }
        if gathered is not None:
            line["all_gather_results"] = gathered
        line.update(post)
        line.update(extras)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(det_w, s, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    det.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the detection hot path on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the detector (stem .. sigmoid, with the fused binarize) over
one batch of 32 synthetic 640x640 frames that are already resident in HBM, per GPU
(weak scaling: every rank owns its own batch and its own replica of the 48.7 MB
weights; there is no data-path collective - results are gathered once, after the
timed region, to show the exchange step).  Prints ONE JSON line on rank 0.

`python bench.py --gpus N` is a complete command for any N: when it is not already running
under torch.distributed.run (no RANK in the environment) and N > 1, this process only LAUNCHES -
before it has made any GPU call it starts N child ranks of itself (subprocess, one per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's JSON line and
exits non-zero if any rank failed.  Under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` the ranks exist already and it just runs as one of them.

Extra objects in that line:
  roofline      dominant kernel (largest summed time), measured live with HIP events
                on the launch stream: the MFMA FLOPs its launches EXECUTE / their time,
                against the dense MFMA peak of the instruction it issues - 157.3 TF/s for the f32 forms,
                2.5 PF/s for v_mfma_f32_32x32x16_bf16, which the split-bf16 kernels (`*_x3*`) issue six times
                per f32 product (frac <= 1 by construction; `algorithmic_tflops` is the reference graph's
                direct-conv work / the same time).  Counter evidence (rocprofv3 --pmc, committed extract
                profiles/pmc.json, stamped with a hash of the kernel sources): HBM traffic, MFMA FLOPs and
                MFMA-busy share per kernel; `*_stale` says when the sources changed after the passes.
  f32_mfma_only the same step with option mfma=f32 (every conv on the exact-f32 matrix instructions).
  host_*        the host-memory entry points (frames in host memory, PCIe inside the timed region).
  roofline_rec  the recogniser kernel, same definition, at B = 256 (configs[2]) and B = 65536.
  cpu_baseline  the same graphs on the host cores through ATen-CPU (oracle/torch_ref.py,
                the operator library the reference reaches through tch), bounded samples:
                detector at all cores (headline) and one thread, recogniser, post-processing.
  N > 1 only    rccl_ranks, all_gather_results_ms, detect_postprocess_gather_images_per_s (forward +
                get_boxes_and_box_scores + the result all-gather inside the timed loop).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, "Chip-level parameters": Peak FP32 (matrix) 157.3 TFLOPS (spec; 155 measured),
# Peak BF16/FP16 MFMA ~2.5 PF dense (the 5 PF vendor figure includes 2:1 sparsity), HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
GFLOP_PER_640_IMAGE = 48.365568  # SURVEY.md section 8(d) / BASELINE.md section 2
MFLOP_PER_CROP = 8.587264


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline detector leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--det-options", default="", help="extra engine options of the headline detector, e.g. overlap=0 (profiling aid; the headline is the default engine)")
    ap.add_argument("--no-extras", action="store_true", help="skip recognition / post-processing side numbers")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only, no GPU: the ranks rendezvous on gloo, all-gather fake polygon lists and rank 0 "
                         "prints a line marked dry_run (used by the CPU test of the launcher path; never a measurement)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the reference's arithmetic (BASELINE configs[1], the headline); bf16 = the opt-in "
                         "OCR_PRECISION_BF16 detector (configs[4]), reported as its own line")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no RANK in the environment.  Makes NO GPU call and imports
# neither torch nor the library; the ranks are child processes (never a re-exec of this one).
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# --------------------------------------------------------------------------------------------------
# CPU placement of the ranks.  N ranks share one host: each rank's post-processing pool, staging threads and ATen pool stay on ITS
# share of the cores - the ones nearest its GPU's NUMA node - instead of floating over both sockets.  Everything here reads sysfs
# only (no GPU call): every rank computes the same plan and takes its own row, whichever launcher started it.
def parse_cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus(sysfs: str = "/sys"):
    """Per GPU, in KFD topology order (the order HIP enumerates in), the CPUs of the NUMA node its PCIe device hangs off
    (sysfs local_cpulist), or None where sysfs does not say.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES lists of indices are applied."""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((d for d in os.listdir(base) if d.isdigit()), key=int)
    except OSError:
        return None
    gpus = []
    for d in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, d, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue   # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
            cpus = parse_cpulist(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "local_cpulist")).read())
            gpus.append(set(cpus) if cpus else None)
        except (OSError, ValueError):
            gpus.append(None)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        v = os.environ.get(var, "")
        if v and all(t.strip().isdigit() for t in v.split(",")):
            gpus = [gpus[int(t)] for t in v.split(",") if int(t) < len(gpus)]
    return gpus or None


def plan_affinity(world: int, allowed, per_rank: int, gpu_cpus=None):
    """`world` disjoint CPU lists of (up to) `per_rank` CPUs out of `allowed`: rank r first takes free CPUs of its GPU's NUMA node
    (gpu_cpus[r], see gpu_local_cpus), then - or when the topology is unknown - the next free ones in order (contiguous slices)."""
    allowed = sorted(allowed)
    per_rank = max(1, min(per_rank, len(allowed) // max(1, world))) if len(allowed) >= world else 1
    free = list(allowed)
    plan = []
    for r in range(world):
        near = gpu_cpus[r] if gpu_cpus and r < len(gpu_cpus) and gpu_cpus[r] else None
        take = [c for c in free if near is None or c in near][:per_rank]
        if len(take) < per_rank:
            take += [c for c in free if c not in take][:per_rank - len(take)]
        if not take:   # more ranks than CPUs: share
            take = [allowed[r % len(allowed)]]
        free = [c for c in free if c not in take]
        plan.append(take)
    return plan


def pin_rank(local: int, world: int, per_rank: int):
    """This process onto its row of the plan (before torch, the library or any thread pool exists).  Returns the CPUs, or None where
    the platform has no sched_setaffinity / the call is refused (the rank then runs unpinned, as before)."""
    if world <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("OCR_BENCH_NO_PIN") == "1":
        return None
    try:
        plan = plan_affinity(world, os.sched_getaffinity(0), per_rank, gpu_local_cpus())
        os.sched_setaffinity(0, plan[local % world])
        return plan[local % world]
    except OSError:
        return None


def launch_ranks(n: int, argv) -> int:
    import tempfile
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            # N ranks share one host: keep each rank's OpenMP / ATen pool small (torch.distributed.run does the same); the
            # library's own post-processing pool is capped at 16 threads per detector
            env.setdefault("OMP_NUM_THREADS", "4")
            # ... and tell each rank its share of the host cores: it sizes the detector's post-processing pool with it
            # (engine option post_threads) instead of every rank starting a pool for the whole machine
            env.setdefault("OCR_BENCH_CORES_PER_RANK", str(max(1, host_cores() // n)))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies leaves the others waiting in a rendezvous or a collective: end them (these exact
        # children, by PID) as soon as one has failed
        failed = False
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                failed = True
                time.sleep(2.0)
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        out0.seek(0)
        text = out0.read()
    sys.stdout.write(text)
    sys.stdout.flush()
    if failed or any(codes):
        sys.stderr.write(f"bench.py: ranks failed, exit codes by rank: {codes}\n")
        return 1
    if not any(line.startswith("{") for line in text.splitlines()):
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    return 0


# --------------------------------------------------------------------------------------------------
def csrc_hash() -> str:
    """sha256 over the kernel sources (ocr-rs_amd/csrc): what a PMC extract is stamped with (no .git on the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ocr-rs_amd", "csrc", "*"))):
        if os.path.isfile(f) and f.rsplit(".", 1)[-1] in ("hip", "hpp", "cpp", "map") or os.path.basename(f) == "Makefile":
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_extract(dtype: str, n: int, s: int):
    """profiles/pmc.json: per-kernel counters of separate rocprofv3 --pmc passes over tools/profile_layers.py
    (tools/make_pmc_json.py).  Returns (kernels dict, source text, stale flag) or (None, None, None)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))[dtype]
        if t.get("batch") == n and t.get("size") == s:
            return t["kernels"], t["source"], t.get("csrc_sha") != csrc_hash()
    except Exception:
        pass
    return None, None, None


PINNED_CPUS = None   # main(): the CPUs this rank pinned itself to (N > 1)
def in_schedule_ms(name: str):
    """Average duration of a kernel's launches in the SHIPPED two-stream schedule, from the committed rocprofv3 pass over the headline
    command (profiles/pmc.json, `in_schedule`; tools/publish_profiles.sh): `avg_launch_ms` is measured live on ONE stream."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))["f32"]["in_schedule"]["kernels"][name]["avg_ms"]
    except Exception:
        return None


STEM_FRAMES_BF16_EXACT = True  # main() clears it when the bench frames are not bf16-exact (they are raw luma: integers 0..255)


def kernel_peak(name: str, dtype: str):
    """(matrix instruction, executed-FLOP multiplier, dense peak TFLOP/s) of a launch label."""
    if name.startswith("stem_x3") and STEM_FRAMES_BF16_EXACT:
        # the benchmark's frames are raw luma (integers 0..255: exactly bf16), so every tile takes the stem's exact fast path:
        # three of the six products have an all-zero operand plane and are skipped (bit-identical sum, stem_tail.hip)
        return "v_mfma_f32_32x32x16_bf16 x3 (raw-luma pixels are exact in bf16; weights split three ways)", 3.0, BF16_MFMA_PEAK_TFLOPS
    if "_x3" in name:
        return "v_mfma_f32_32x32x16_bf16 x6 (3-way split f32 operands)", 6.0, BF16_MFMA_PEAK_TFLOPS
    if dtype == "bf16" and ("bf16" in name or name.startswith(("stem_", "tail_"))):
        return "v_mfma_f32_32x32x16_bf16", 1.0, BF16_MFMA_PEAK_TFLOPS
    return "v_mfma_f32 (f32 operands)", 1.0, F32_MFMA_PEAK_TFLOPS


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


def cpu_baseline(det_w, rec_w, size: int, budget_s: float):
    """Reference stand-in on the host cores: the same graphs through ATen CPU kernels
    (oracle/torch_ref.py), and the post-processing restatement (oracle/postproc_oracle.py).  Bounded samples."""
    import numpy as np
    import torch
    from oracle import torch_ref as T
    from ocr_rs_amd import weights as W
    from tests import fixtures as FX
    cores = host_cores()

    def det_rate(threads, batch, max_frames, budget):
        torch.set_num_threads(threads)
        x = W.synth_image_batch(1, batch, size, size)
        T.det_forward(det_w, x[:1])  # warm the thread pool / allocator
        done, t0 = 0, time.perf_counter()
        while True:
            T.det_forward(det_w, x)
            done += x.shape[0]
            el = time.perf_counter() - t0
            if el > budget or done >= max_frames:
                return done, el

    done, el = det_rate(cores, 2, 64, budget_s)
    out = {"value": round(done / el, 3), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{done} frames {size}x{size}, oracle/torch_ref.py (ATen CPU), {cores} threads, {el:.1f} s"}
    d1, e1 = det_rate(1, 1, 2, 4.0)
    out["one_thread"] = {"value": round(d1 / e1, 3), "unit": "images/s", "cores": 1, "sample": f"{d1} frames, {e1:.1f} s"}
    torch.set_num_threads(cores)
    crops = W.synth_crops(2, 4096)
    T.rec_forward(rec_w, crops[:256])
    t0, it = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        T.rec_classify(T.rec_forward(rec_w, crops))
        it += 1
    el = time.perf_counter() - t0
    out["recognition"] = {"value": round(it * 4096 / el, 1), "unit": "crops/s", "cores": cores, "kind": "port",
                          "sample": f"{it} x 4096 crops 28x28, oracle/torch_ref.py, {el:.1f} s"}
    # post-processing: the compiled CPU path (oracle/postproc_cpu.cpp: binarize, contours, Douglas-Peucker, box scores,
    # unclip - all on host threads; pinned to the reference's known answers by tests/test_oracle_postproc.py), one thread
    # and all cores, on the text-like maps the GPU leg uses
    from oracle import postproc_cpu as PC
    maps = FX.text_like_maps(32, size, 7)
    ones = np.ones((32, 2))
    PC.get_boxes_and_box_scores(maps[:2], ones[:2], threads=1, skip_degenerate=True, counts_only=True)

    def post_rate(threads, budget):
        t0, it = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            PC.get_boxes_and_box_scores(maps, ones, threads=threads, skip_degenerate=True, counts_only=True)
            it += 1
        return it, time.perf_counter() - t0

    it, el = post_rate(cores, 2.0)
    out["postprocess"] = {"value": round(it * 32 / el, 1), "unit": "images/s", "cores": cores, "kind": "port",
                          "sample": f"{it} x 32 text-like maps, oracle/postproc_cpu.cpp (product's host geometry, DESIGN 6), {el:.1f} s"}
    it1, el1 = post_rate(1, 1.5)
    out["postprocess"]["one_thread"] = {"value": round(it1 * 32 / el1, 1), "unit": "images/s", "cores": 1, "sample": f"{it1} x 32 maps, {el1:.1f} s"}
    # ... and an INDEPENDENT one: the pure-Python restatement (oracle/postproc_oracle.py) - none of the product's code, one interpreter
    # thread, a handful of maps.  (The compiled path above shares its geometry source with the library; this one shares nothing.)
    from oracle import postproc_oracle as O
    t0, k = time.perf_counter(), 0
    while k < 64 and time.perf_counter() - t0 < 4.0:
        O.get_boxes_and_box_scores(maps[k % 32:k % 32 + 1], ones[:1], skip_degenerate=True)
        k += 1
    el2 = time.perf_counter() - t0
    out["postprocess"]["python_oracle"] = {"value": round(k / el2, 2), "unit": "images/s", "cores": 1, "kind": "port",
                                           "sample": f"{k} text-like maps, oracle/postproc_oracle.py (pure Python), {el2:.1f} s"}
    return out


def dry_run(a) -> None:
    """The multi-rank control flow without a GPU (gloo): rendezvous, result all-gather, one line from rank 0."""
    import torch
    import torch.distributed as dist
    import ocr_rs_amd  # noqa: F401
    from ocr_rs_amd import parallel as P
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("OCR_BENCH_FAIL_RANK") == str(rank):
        sys.exit(3)  # test hook: a rank that dies must fail the launcher
    images = 0
    masks = None
    if world > 1:
        dist.init_process_group("gloo")
        polys = [[[(rank, i), (rank + 1, i), (rank + 1, i + 1), (rank, i + 1)]] * (i % 3) for i in range(a.batch)]
        scores = [[0.9] * (i % 3) for i in range(a.batch)]
        all_p, _ = P.all_gather_results(polys, scores, torch.device("cpu"))
        images = len(all_p)
        masks = [None] * world   # every rank's CPU mask as it stands after pin_rank (what the GPU run uses)
        dist.all_gather_object(masks, sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = {"dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "all_gather_results": {"images": images},
                "cores_per_rank": int(os.environ.get("OCR_BENCH_CORES_PER_RANK") or max(1, host_cores() // world)),
                "rank_cpus": masks if world > 1 else None, "pinned": PINNED_CPUS is not None}
        if os.environ.get("OCR_BENCH_FAIL_EXCHANGE") == "1":   # test hook: the path a stalled / failed C-ABI exchange takes
            line["exchange_ok"] = False
            emit(line)
            sys.exit(4)
        emit(line)


def c_abi_exchange(capi, dist, torch, polys, scores, gathered, world, rank, local, line, deadline_s=90.0):
    """The result exchange through the C ABI (ocr_comm_*: RCCL called from the library, no torch in the data path) - what a
    non-Python host binds.  It runs LAST, under a deadline: if a rank fails or stalls inside the library's communicator, every
    rank leaves through the timer (rank 0 prints the line it already has, with the error noted) instead of hanging the job."""
    import threading

    def expire():   # a stalled communicator must never read as success: the line says so and the rank exits non-zero
        if line is not None:
            line["all_gather_results_c_abi_error"] = f"no completion within {deadline_s:.0f} s"
            line["exchange_ok"] = False
            emit(line)
        os._exit(3)

    dist.barrier()  # rank 0 arrives last (it ran the extras): the deadline counts from the moment everyone is here
    timer = threading.Timer(deadline_s, expire)
    timer.daemon = True
    timer.start()
    out = {}
    try:
        idb = [capi.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(idb, src=0)
        comm = capi.Comm(idb[0], world, rank, local)
        cp, cs = comm.all_gather_polygons(polys, scores)
        same = cp == gathered[0] and cs == gathered[1]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            comm.all_gather_polygons(polys, scores)
        ms = (time.perf_counter() - t1) / 10 * 1e3
        out = {"exchange_ok": bool(same), "all_gather_results_c_abi_ms": round(ms, 3), "all_gather_results_c_abi_matches": bool(same),
               "all_gather_results_c_abi_note": "rank 0's wall time per call, ocr_comm_all_gather_polygons",
               "rccl_version_c_abi": capi.Comm.rccl_version()}
        comm.close()
    except Exception as e:
        out = {"exchange_ok": False, "all_gather_results_c_abi_error": f"{type(e).__name__}: {e}"}
    timer.cancel()
    return out


_RESULT_FD = None


def reserve_stdout() -> None:
    """Keep the process's stdout for the ONE result line: libraries write banners there (RCCL prints its version block to
    stdout when a communicator is created, gloo its connection note), so fd 1 is pointed at stderr for everything else."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def emit(line) -> None:
    data = (json.dumps(line) + "\n").encode()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, data)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    reserve_stdout()
    # a rank of N > 1 first moves onto its share of the host cores (nearest its GPU's NUMA node): before torch, the library or any pool
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    global PINNED_CPUS
    # (the share is fixed BEFORE pinning - host_cores() reads the affinity mask - and kept in the environment for the later readers)
    os.environ.setdefault("OCR_BENCH_CORES_PER_RANK", str(max(1, host_cores() // max(1, world_env))))
    PINNED_CPUS = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE") or world_env),
                           int(os.environ["OCR_BENCH_CORES_PER_RANK"]))
    if a.dry_run:
        return dry_run(a)

    import numpy as np
    import torch

    import ocr_rs_amd  # noqa: F401
    from ocr_rs_amd import capi
    from ocr_rs_amd import weights as W
    from tests import fixtures as FX

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one rank per GPU; OCR_BENCH_BACKEND=gloo (+ fewer GPUs than ranks) only exists to rehearse the
    # multi-rank control flow on a 1-GPU box
    backend = os.environ.get("OCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    # OCR_BENCH_C_ABI=1 with the gloo rehearsal: attempt the library's own RCCL communicator anyway (two ranks on one GPU make
    # RCCL refuse - that exercises the error path of c_abi_exchange)
    try_c_abi = backend == "nccl" or os.environ.get("OCR_BENCH_C_ABI") == "1"
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
    comm_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")

    n, s = a.batch, a.size
    det_w = W.make_det_weights(0)
    # host threads of this rank's post-processing pool: its share of the cores when several ranks share the host
    cores_per_rank = int(os.environ.get("OCR_BENCH_CORES_PER_RANK") or max(1, host_cores() // world))
    det_opts = f"post_threads={min(16, cores_per_rank)}" + (";" + a.det_options if a.det_options else "")
    det = capi.Detector(W.pack_blob(det_w), local, options=det_opts)
    if a.dtype == "bf16":
        det.set_precision(capi.PRECISION_BF16)
    stream = torch.cuda.Stream(device=local)
    det.set_stream(stream.cuda_stream)
    x_np = W.synth_image_batch(1 + rank, n, s, s)
    global STEM_FRAMES_BF16_EXACT
    # the stem skips three of its six products per tile only when the tile's pixels are exact in bf16 (stem_tail.hip): price it
    # at 3 products only when that holds for the frames actually benchmarked
    STEM_FRAMES_BF16_EXACT = bool((x_np == np.round(x_np)).all() and x_np.min() >= 0 and x_np.max() <= 256)
    x = torch.from_numpy(x_np).to(f"cuda:{local}")
    prob = torch.empty_like(x)
    bitmap = torch.empty(x.shape, dtype=torch.uint8, device=x.device)

    def step():
        det.forward_device(x.data_ptr(), n, s, s, prob.data_ptr(), bitmap.data_ptr(), 0.6)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v: float) -> float:
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    with torch.cuda.stream(stream):
        for _ in range(a.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed)

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
        # the engine reports the f32 MULTIPLY-ADDS a launch executes (2 per MAC): on the f32 matrix instructions those are
        # its MFMA FLOPs; a split-bf16 launch issues six bf16 MFMA products per f32 one.  `achieved` / `frac` are matrix-core
        # utilisation against the peak of the instruction issued.  A fused Winograd launch is a 3x3 conv whose direct-algorithm
        # work (2 M N 9C, what SURVEY 8d counts for the reference graph) is 4 times what it executes (F(4x4,3x3)); that is
        # reported beside, never as the fraction.
        instr, mult, kpeak = kernel_peak(name, a.dtype)
        executed = mult * fl / (ms * 1e-3) / 1e12
        alg_factor = 4.0 if name.startswith("winograd43_fused") else 1.0
        executed_gflop = sum(v[1] * kernel_peak(k, a.dtype)[1] for k, v in agg.items()) / reps / 1e9
        pmc, pmc_source, pmc_stale = pmc_extract(a.dtype, n, s)
        pk = (pmc or {}).get(name, {})
        allk = {}
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            ki, km, kp = kernel_peak(k, a.dtype)
            tf = km * v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else None
            e = {"ms_per_step": round(v[0] / reps, 4), "launches_per_step": v[3] // reps, "tflops": None if tf is None else round(tf, 2),
                 "peak": kp, "frac": None if not tf else round(tf / kp, 4), "gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else None}
            c = (pmc or {}).get(k)
            if c:   # counters per launch: HBM bytes, MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_* x 512), MFMA-busy share of the launch
                e["pmc"] = {kk: c[kk] for kk in ("hbm_bytes", "mfma_flops", "mfma_busy") if kk in c}
                if c.get("hbm_bytes") and v[0] > 0:
                    e["pmc"]["hbm_frac"] = round(c["hbm_bytes"] / (v[0] / v[3] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            allk[k] = e
        roof = {"kernel": name, "bound": "mfma", "instruction": instr, "achieved": round(executed, 2), "peak": kpeak,
                "unit": "TFLOP/s", "frac": round(executed / kpeak, 4),
                "definition": "MFMA FLOPs the kernel executes per launch / average launch duration (HIP events on the launch stream)",
                "algorithmic_tflops": round(executed * alg_factor, 2), "algorithmic_speedup": round(alg_factor, 3),
                "traffic": pk.get("hbm_bytes"), "traffic_unit": "HBM bytes per launch", "traffic_source": pmc_source,
                "traffic_stale": pmc_stale,
                "mfma_flops_counter": pk.get("mfma_flops"), "mfma_busy": pk.get("mfma_busy"),
                "counter_note": "per launch, rocprofv3 --pmc in separate passes over tools/profile_layers.py (tools/collect_profiles.sh): "
                                "FETCH_SIZE x 2 (gfx950) + WRITE_SIZE; SQ_INSTS_VALU_MFMA_MOPS_F32/BF16 x 512; "
                                "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE per XCD)",
                "launches_per_step": cnt // reps, "avg_launch_ms": round(ms / cnt, 4),
                "avg_launch_ms_in_schedule": in_schedule_ms(name),
                "avg_launch_note": "one stream (ocr_det_forward_profile).  In the timed steps (engine option overlap=3, default) two of this kernel's six launches per step run on the "
                                   "side stream beside layer3 / layer4 and stretch; rocprofv3's average over ALL launches of the default command is therefore higher "
                                   "(profiles/README.md); with --det-options overlap=0 it is this number",
                "avg_launch_gflop_executed": round(mult * fl / cnt / 1e9, 3),
                "all_kernels": allk}
        if name.startswith("winograd43_fused"):
            # what actually bounds this kernel: its B operand (the transformed weights, 36 matrices per 16
            # input channels) streams from L2 into registers once per 16 x 16 pixel block - 36 x C/16 x 4 KB per wave - next to
            # the input patches, the residual and the stores
            cin = 64 if "c64" in name else 128 if "c128" in name else 256
            blocks = n * ((s // 4 + 15) // 16) ** 2 if cin == 64 else None
            if blocks:
                l2 = blocks * ((cin // 16) * 36 * 4096 + 18 * 18 * cin * 4 + 2 * 256 * 64 * 4)
                roof["l2_to_cu"] = {"bytes_per_launch": l2, "achieved_gbs": round(l2 / (ms / cnt * 1e-3) / 1e9, 1),
                                    "measured_ceiling_gbs": round(70.0 * 256, 1),
                                    "note": "operand bytes that cross L2 -> CU per launch (weight fragments 590 KB per block, patch, residual) / "
                                            "time; ceiling 66-73 GB/s per CU x 256 (MI355X_MICROARCH.md, gather from L2)"}

    # ---- the same step with every conv on the exact-f32 matrix instructions (option mfma=f32): what the split-bf16
    # kernels buy, measured beside the headline in the same process
    strict = {}
    if rank == 0 and a.dtype == "f32" and not a.no_extras:
        try:
            d32 = capi.Detector(W.pack_blob(det_w), local, options="mfma=f32;" + det_opts)
            d32.set_stream(stream.cuda_stream)
            k = max(5, a.steps // 2)
            with torch.cuda.stream(stream):
                for _ in range(2):
                    d32.forward_device(x.data_ptr(), n, s, s, prob.data_ptr(), bitmap.data_ptr(), 0.6)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    d32.forward_device(x.data_ptr(), n, s, s, prob.data_ptr(), bitmap.data_ptr(), 0.6)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
            strict["f32_mfma_only"] = {"images_per_s": round(n * k / el, 1), "ms_per_step": round(el / k * 1e3, 3), "steps": k,
                                       "note": "engine option mfma=f32: no split-bf16 kernels, every conv on v_mfma_f32_32x32x2_f32 / 16x16x4_f32"}
            d32.close()
        except Exception as e:
            strict["f32_mfma_only_error"] = f"{type(e).__name__}: {e}"

    # ---- the opt-in bf16 precision of configs[4], measured in EVERY default run (never the headline): the same detector handle
    # switched to precision=bf16 for ten steps, its dominant kernel from per-launch HIP events, and the whole hot path
    # (detect -> polygons -> crops -> classify) further down (`bf16.e2e_pages_per_s`)
    bf16 = {}
    if rank == 0 and a.dtype == "f32" and not a.no_extras:
        try:
            det.set_precision(capi.PRECISION_BF16)
            k = 10
            with torch.cuda.stream(stream):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    step()
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
                agg = {}
                for _ in range(2):
                    for name, ms, fl, by in det.forward_profile(x.data_ptr(), n, s, s, prob.data_ptr()):
                        e = agg.setdefault(name, [0.0, 0.0, 0])
                        e[0] += ms
                        e[1] += fl
                        e[2] += 1
            dname, (dms, dfl, dcnt) = max(agg.items(), key=lambda kv: kv[1][0])
            _, mult, kpeak = kernel_peak(dname, "bf16")
            dtf = mult * dfl / (dms * 1e-3) / 1e12
            bf16 = {"ms_per_step": round(el / k * 1e3, 3), "images_per_s": round(n * k / el, 1), "steps": k,
                    "dominant_kernel": {"kernel": dname, "avg_launch_ms": round(dms / dcnt, 4), "launches_per_step": dcnt // 2,
                                        "tflops": round(dtf, 1), "peak": kpeak, "frac": round(dtf / kpeak, 4)},
                    "note": "engine precision=bf16 (BASELINE configs[4]): bf16 operands, f32 accumulate; same handle, same frames, after the headline"}
        except Exception as e:
            bf16 = {"error": f"{type(e).__name__}: {e}"}
        finally:
            det.set_precision(capi.PRECISION_F32)

    # ---- frames in HOST memory (the reference's call sites hand CPU tensors): PCIe inside the timed region.  Blocking forward
    # over pipelined pieces (ocr_det_forward / _u8, MEM_HOST) from pinned and from pageable memory.
    host = {}
    if rank == 0 and not a.no_extras and a.dtype == "f32":
        try:
            xf = x.cpu().numpy()
            xb = np.clip(np.rint(xf), 0, 255).astype(np.uint8)
            pin_f, pin_b, pin_p = capi.HostBuffer(xf.shape, np.float32), capi.HostBuffer(xb.shape, np.uint8), capi.HostBuffer(xf.shape, np.float32)
            pin_f.array[...] = xf
            pin_b.array[...] = xb
            page_p = np.empty_like(xf)
            L = capi.lib()

            def rate(fn, reps):
                fn()
                t1 = time.perf_counter()
                for _ in range(reps):
                    fn()
                return round(n * reps / (time.perf_counter() - t1), 1)

            k = max(4, a.steps // 4)
            host["host_forward_images_per_s"] = {
                "pinned_f32": rate(lambda: capi.check(L.ocr_det_forward(det._h, pin_f.array.ctypes.data, n, s, s, pin_p.array.ctypes.data, capi.MEM_HOST)), k),
                "pinned_u8": rate(lambda: capi.check(L.ocr_det_forward_u8(det._h, pin_b.array.ctypes.data, n, s, s, pin_p.array.ctypes.data, capi.MEM_HOST)), k),
                "pageable_f32": rate(lambda: capi.check(L.ocr_det_forward(det._h, xf.ctypes.data, n, s, s, page_p.ctypes.data, capi.MEM_HOST)), k),
                "pageable_u8": rate(lambda: capi.check(L.ocr_det_forward_u8(det._h, xb.ctypes.data, n, s, s, page_p.ctypes.data, capi.MEM_HOST)), k),
                "note": "blocking call, frames and maps in host memory (52 MB f32 / 13 MB u8 in, 52 MB out per batch), copies and forward "
                        "pipelined over four pieces of the batch; the device-resident rate is `value`"}
            for b in (pin_f, pin_b, pin_p):
                b.close()
        except Exception as e:
            host["host_forward_error"] = f"{type(e).__name__}: {e}"

    # ---- post-processing and the exchange step of the sharded path: every rank post-processes text-like maps
    # of its shard (get_boxes_and_box_scores) and the variable-length polygon blocks are all-gathered -
    # RCCL over xGMI when the backend is nccl (ocr-rs_amd/parallel.py)
    post = {}
    polys = scores = pm = adj = params = gathered = None
    try:
        params = capi.default_params(skip_degenerate=True)
        maps = FX.text_like_maps(n, s, seed=rank)
        pm = torch.from_numpy(maps).to(x.device)
        adj = np.ones((n, 2))
        torch.cuda.synchronize()
        polys, scores = det.postprocess(pm, n, s, s, adj, capi.MEM_DEVICE, params)
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):   # the library call (ocr_det_postprocess), not the conversion of its CSR block to Python lists
            det.postprocess_counts(pm, n, s, s, adj, capi.MEM_DEVICE, params)
        post = {"postprocess_images_per_s": round(n * reps / (time.perf_counter() - t1), 1),
                "postprocess_polygons_per_image": round(sum(len(p) for p in polys) / n, 2)}
        if rank == 0 and not a.no_extras:
            dm = torch.from_numpy(FX.dense_text_maps(n, s, 5)).to(x.device)
            npoly, _ = det.postprocess_counts(dm, n, s, s, adj, capi.MEM_DEVICE, params)
            t1 = time.perf_counter()
            for _ in range(reps):
                det.postprocess_counts(dm, n, s, s, adj, capi.MEM_DEVICE, params)
            post["postprocess_dense_images_per_s"] = round(n * reps / (time.perf_counter() - t1), 1)
            post["postprocess_dense_polygons_per_image"] = round(npoly / n, 2)
    except Exception as e:  # side numbers never hide the headline
        post["postprocess_error"] = f"{type(e).__name__}: {e}"
        polys = None
    if dist is not None:
        # the collective section runs only when EVERY rank has results (a rank that failed above must not leave
        # the others waiting in an all-gather)
        ok = torch.tensor([0.0 if polys is None else 1.0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            from ocr_rs_amd import parallel as P
            all_p, all_s = P.all_gather_results(polys, scores, comm_dev)
            fence()
            t1 = time.perf_counter()
            for _ in range(10):
                P.all_gather_results(polys, scores, comm_dev)
            gather_ms = max_over_ranks((time.perf_counter() - t1) / 10 * 1e3)
            # configs[3] as a pipeline: forward + get_boxes_and_box_scores + all-gather of the polygon lists per step
            with torch.cuda.stream(stream):
                fence()
                k = max(3, a.steps // 4)
                t1 = time.perf_counter()
                for _ in range(k):
                    step()
                    pl, sc = det.postprocess(pm, n, s, s, adj, capi.MEM_DEVICE, params)
                    P.all_gather_results(pl, sc, comm_dev)
                fence()
                e2e = max_over_ranks(time.perf_counter() - t1)
            gathered = (all_p, all_s)
            post.update({"rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                         "collective_backend": backend + (f" (RCCL {'.'.join(map(str, torch.cuda.nccl.version()))})" if backend == "nccl" else ""),
                         "all_gather_results_ms": round(gather_ms, 3),
                         "all_gather_results": {"images": len(all_p), "polygons": sum(len(p) for p in all_p)},
                         "detect_postprocess_gather_images_per_s": round(n * world * k / e2e, 1),
                         "detect_postprocess_gather_note": "per step and rank: forward of its 32 frames, get_boxes_and_box_scores over 32 "
                                                           "text-like maps, all-gather of the polygon lists; sequential on one stream"})

    # ---- configs[3] end to end at its best: every rank runs the library's pipelined detect (forward of batch k+1 under the
    # post-processing of batch k) on synthetic pages with text-following weights, and the polygon lists of the batch that just
    # came back are all-gathered each step.  Set-up may fail on a rank; the collective loop runs only if it worked on all.
    if dist is not None and gathered is not None and not a.no_extras:
        dt = xp = None
        try:
            dt = capi.Detector(W.pack_blob(W.make_det_weights_text()), local, options=det_opts)
            dt.set_stream(stream.cuda_stream)
            pages, _ = W.synth_text_pages(77 + rank, n, s, s)
            xp = torch.from_numpy(pages).to(x.device)
            pr2 = [torch.empty_like(xp), torch.empty_like(xp)]
            adj1 = np.ones((n, 2))
            params = capi.default_params(skip_degenerate=True)
            torch.cuda.synchronize()
        except Exception as e:
            post["detect_pipelined_gather_error"] = f"{type(e).__name__}: {e}"
            dt = None
        ok2 = torch.tensor([0.0 if dt is None else 1.0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
        if float(ok2.item()) > 0:
            k = max(4, a.steps // 4)
            images = 0
            for it in range(2):                      # warm-up, then timed
                fence()
                t1 = time.perf_counter()
                for j in range(k + 1):               # k batches in, one flush: k result blocks come back
                    r = (dt.detect_pipelined(xp.data_ptr(), n, s, s, pr2[j & 1].data_ptr(), adj1, params) if j < k
                         else dt.detect_pipelined(0, 0, 0, 0, 0))
                    if r is not None:
                        ap, _ = P.all_gather_results(r[0], r[1], comm_dev)
                        images = len(ap)
                fence()
                e2e = max_over_ranks(time.perf_counter() - t1)
            post["detect_pipelined_gather_images_per_s"] = round(n * world * k / e2e, 1)
            post["detect_pipelined_gather_note"] = ("per rank: ocr_det_detect_pipelined over its 32 synthetic pages (forward of batch k+1 under "
                                                    "the post-processing of batch k), all-gather of each returned polygon block "
                                                    f"({images} images per gather)")
        if dt is not None:
            dt.close()

    # ---- detection END TO END on one GPU: forward + get_boxes_and_box_scores of the forward's own maps, software-
    # pipelined inside the library (ocr_det_detect_pipelined).  Random weights give noise maps, so this leg runs the
    # text-following synthetic weights on synthetic pages (weights.make_det_weights_text / synth_text_pages): same graph,
    # same kernels, maps with ~20 word polygons per page.
    if rank == 0 and not a.no_extras and a.dtype == "f32":
        try:
            dt = capi.Detector(W.pack_blob(W.make_det_weights_text()), local, options=det_opts)
            dt.set_stream(stream.cuda_stream)
            pages, boxes = W.synth_text_pages(77, n, s, s)
            xp = torch.from_numpy(pages).to(x.device)
            pr2 = [torch.empty_like(xp), torch.empty_like(xp)]
            adj1 = np.ones((n, 2))
            params = capi.default_params(skip_degenerate=True)
            torch.cuda.synchronize()
            k, found = max(6, a.steps // 2), 0
            for it in range(2):                      # warm-up, then timed
                t1 = time.perf_counter()
                for j in range(k):
                    r = dt.detect_pipelined(xp.data_ptr(), n, s, s, pr2[j & 1].data_ptr(), adj1, params, convert=False)
                    found += r[0] if (r and it) else 0
                r = dt.detect_pipelined(0, 0, 0, 0, 0, convert=False)
                found += r[0] if it else 0
                torch.cuda.synchronize()
                e2e = time.perf_counter() - t1
            post["detect_postprocess_pipelined_images_per_s"] = round(n * k / e2e, 1)
            # the same stream of batches from HOST memory: pinned u8 / f32 frames, polygons back, maps stay on the device
            pb = np.clip(np.rint(pages), 0, 255).astype(np.uint8)
            hp = {}
            for label, arr in (("pinned_u8", pb), ("pinned_f32", pb.astype(np.float32)), ("pageable_u8", pb)):
                bufs = []
                for j in range(2):
                    if label.startswith("pinned"):
                        hb = capi.HostBuffer(arr.shape, arr.dtype)
                        hb.array[...] = arr
                        bufs.append(hb)
                    else:
                        bufs.append(None)
                for it in range(2):
                    t1 = time.perf_counter()
                    for j in range(k):
                        src = bufs[j & 1].array if bufs[j & 1] is not None else arr
                        dt.detect_pipelined_host(src, adjust_values=adj1, params=params, convert=False)
                    dt.detect_pipelined_host(None)
                    e2h = time.perf_counter() - t1
                hp[label] = round(n * k / e2h, 1)
                for hb in bufs:
                    if hb is not None:
                        hb.close()
            hp["note"] = ("ocr_det_detect_pipelined_host: per batch the frames cross PCIe into the staging slots beside the previous forward, "
                          "polygons come back, the probability map stays on the device")
            post["host_to_polygons_images_per_s"] = hp
            post["detect_postprocess_pipelined_polygons_per_image"] = round(found / (n * k), 2)
            post["detect_postprocess_pipelined_note"] = ("ocr_det_detect_pipelined: forward of batch k+1 overlapped with binarize + contours + "
                                                         "box scores + unclip of batch k; text-following synthetic weights and pages")
            dt.close()
        except Exception as e:
            post["detect_postprocess_pipelined_error"] = f"{type(e).__name__}: {e}"

    # ---- configs[4] / the whole hot path on one GPU: pages -> detect (pipelined with its post-processing) -> crops of every
    # polygon -> classify, in this run's precision; 128 synthetic pages in four batches of 32
    e2e = {}
    if rank == 0 and not a.no_extras:
        def e2e_pages(dtype):
            de = capi.Detector(W.pack_blob(W.make_det_weights_text()), local, options=det_opts)
            if dtype == "bf16":
                de.set_precision(capi.PRECISION_BF16)
            de.set_stream(stream.cuda_stream)
            re_ = capi.Recognizer(W.pack_blob(W.make_rec_weights(0)), local)
            rec_stream = torch.cuda.Stream(device=local)     # the recogniser beside the next forward, not behind it
            re_.set_stream(rec_stream.cuda_stream)
            nb = 4
            xs = [torch.from_numpy(W.synth_text_pages(300 + j, n, s, s)[0]).to(x.device) for j in range(nb)]
            prs = [torch.empty_like(xs[0]), torch.empty_like(xs[0])]
            cap = 4096
            crops = torch.empty((cap, 784), dtype=torch.float32, device=x.device)
            labels = torch.empty(cap, dtype=torch.int32, device=x.device)
            probs_c = torch.empty(cap, dtype=torch.float64, device=x.device)
            adj1 = np.ones((n, 2))
            params = capi.default_params(skip_degenerate=True)
            torch.cuda.synchronize()
            ncrops = 0
            for it in range(2):                      # warm-up, then timed
                t1 = time.perf_counter()
                ncrops = 0
                for j in range(nb + 1):
                    blk = (de.detect_pipelined_block(xs[j].data_ptr(), n, s, s, prs[j & 1].data_ptr(), adj1, params) if j < nb
                           else de.detect_pipelined_block(0, 0, 0, 0, 0))
                    if blk is not None:
                        if blk.contents.n_polygons > cap:
                            raise RuntimeError(f"{blk.contents.n_polygons} crops in one batch")
                        k = de.extract_crops_block(blk, xs[j - 1].data_ptr(), n, s, s, adj1, crops.data_ptr())
                        if k:
                            re_.classify_device(crops.data_ptr(), k, 0, labels.data_ptr(), probs_c.data_ptr())
                            re_.synchronize()
                        ncrops += k
                        de.free_block(blk)
                torch.cuda.synchronize()
                el = time.perf_counter() - t1
            de.close()
            re_.close()
            return round(n * nb / el, 1), round(ncrops / (n * nb), 2), n * nb

        try:
            pps, cpp, npages = e2e_pages(a.dtype)
            e2e = {"e2e_pages_per_s": pps, "e2e_crops_per_page": cpp, "e2e_pages": npages,
                   "e2e_note": f"detect (ocr_det_detect_pipelined, {a.dtype}) -> polygons -> ocr_extract_crops -> ocr_rec_classify on synthetic pages with "
                               "text-following weights, device-resident frames, crops and labels"}
        except Exception as e:
            e2e = {"e2e_error": f"{type(e).__name__}: {e}"}
        if a.dtype == "f32" and "error" not in bf16 and bf16:
            try:
                bf16["e2e_pages_per_s"], bf16["e2e_crops_per_page"], bf16["e2e_pages"] = e2e_pages("bf16")
            except Exception as e:
                bf16["e2e_error"] = f"{type(e).__name__}: {e}"

    # ---- how many host cores one detector's post-processing needs (8 ranks share a host: cores_per_rank): the pipelined detect
    # from device-resident frames and from pinned host memory with the handle's pool at 1, 2, 4 and 16 threads, on text-like
    # pages (18 words) and dense ones (about 65 words per page), in both precisions
    sweep = {}
    if rank == 0 and not a.no_extras and a.dtype == "f32":
        try:
            params = capi.default_params(skip_degenerate=True)
            adj1 = np.ones((n, 2))
            pages = {"text": W.synth_text_pages(77, n, s, s)[0], "dense": W.synth_text_pages(78, n, s, s, dense=True)[0]}
            dev = {k2: torch.from_numpy(v).to(x.device) for k2, v in pages.items()}
            pin = {}
            for k2, v in pages.items():
                hb = capi.HostBuffer(v.shape, np.uint8)
                hb.array[...] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
                pin[k2] = hb
            pr2 = [torch.empty_like(dev["text"]), torch.empty_like(dev["text"])]
            kk = 6
            # (threads, options, key): the engine's default placement of the polygon chain per pool size, and at 16 threads the whole
            # chain forced onto the device (what device_contours=auto picks for small pools) beside it
            for threads, extra, key in ((1, "", "1"), (2, "", "2"), (4, "", "4"), (16, "", "16"),
                                        (16, ";device_contours=1;device_unclip=2", "16_device_chain")):
                dt = capi.Detector(W.pack_blob(W.make_det_weights_text()), local,
                                   options=f"post_threads={threads}" + extra + (";" + a.det_options if a.det_options else ""))
                dt.set_stream(stream.cuda_stream)
                row = {}
                for prec in ("f32", "bf16"):
                    dt.set_precision(capi.PRECISION_BF16 if prec == "bf16" else capi.PRECISION_F32)
                    for kind in ("text", "dense"):
                        found, el_d, el_h = 0, float("inf"), float("inf")
                        for it in range(3):                      # warm-up, then the better of two timed passes
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            got = 0
                            for j in range(kk):
                                r = dt.detect_pipelined(dev[kind].data_ptr(), n, s, s, pr2[j & 1].data_ptr(), adj1, params, convert=False)
                                got += r[0] if r else 0
                            r = dt.detect_pipelined(0, 0, 0, 0, 0, convert=False)
                            got += r[0]
                            torch.cuda.synchronize()
                            if it:
                                el_d = min(el_d, time.perf_counter() - t1)
                                found = got
                        for it in range(3):
                            t1 = time.perf_counter()
                            for j in range(kk):
                                dt.detect_pipelined_host(pin[kind].array, adjust_values=adj1, params=params, convert=False)
                            dt.detect_pipelined_host(None)
                            if it:
                                el_h = min(el_h, time.perf_counter() - t1)
                        row[f"{prec}_{kind}"] = {"detect_postprocess_pipelined_images_per_s": round(n * kk / el_d, 1),
                                                 "host_to_polygons_images_per_s": round(n * kk / el_h, 1),
                                                 "polygons_per_image": round(found / (n * kk), 1)}
                row["where"] = dt.post_stats()
                sweep[key] = row
                dt.close()
            for hb in pin.values():
                hb.close()
            sweep["note"] = ("ocr_det_detect_pipelined (device frames) and ocr_det_detect_pipelined_host (pinned u8 frames) with the handle's "
                             "post-processing pool at 1 / 2 / 4 / 16 threads; text: 18 words per 640 x 640 page, dense: about 65")
        except Exception as e:
            sweep["error"] = f"{type(e).__name__}: {e}"

    extras = {}
    rec_w = W.make_rec_weights(0)
    if rank == 0 and not a.no_extras:
        try:
            rec = capi.Recognizer(W.pack_blob(rec_w), local)
            rec.set_stream(stream.cuda_stream)
            roof_rec = {"bound": "mfma", "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "definition": "MFMA FLOPs a launch executes / its duration (HIP events on the launch stream); the dominant "
                                      "kernel of the pass is the one named; whole_pass = 8.587264 MFLOP per crop (SURVEY Appendix C) "
                                      "x crops / device time of all launches of the pass"}
            for nc in (256, 65536):
                crops = torch.from_numpy(W.synth_crops(2, nc)).to(x.device)
                labels = torch.empty(nc, dtype=torch.int32, device=x.device)
                probs = torch.empty(nc, dtype=torch.float64, device=x.device)
                it = 50 if nc == 256 else 5
                with torch.cuda.stream(stream):
                    for _ in range(3):
                        rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    e0.record(stream)
                    for _ in range(it):
                        rec.classify_device(crops.data_ptr(), nc, 0, labels.data_ptr(), probs.data_ptr())
                    e1.record(stream)
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t1
                    ragg = {}
                    for _ in range(3):
                        for name, ms, fl, by in rec.classify_profile(crops.data_ptr(), nc, labels.data_ptr(), probs.data_ptr()):
                            e = ragg.setdefault(name, [0.0, 0.0, 0])
                            e[0] += ms
                            e[1] += fl
                            e[2] += 1
                ms = e0.elapsed_time(e1) / it
                tf = MFLOP_PER_CROP * 1e6 * nc / (ms * 1e-3) / 1e12
                extras[f"rec_crops_per_s_b{nc}"] = round(nc * it / wall, 1)
                dname, (dms, dfl, dcnt) = max(ragg.items(), key=lambda kv: kv[1][0])
                dtf = dfl / (dms * 1e-3) / 1e12
                roof_rec[f"b{nc}"] = {"kernel": dname, "avg_launch_ms": round(dms / dcnt, 4), "achieved": round(dtf, 2),
                                      "frac": round(dtf / F32_MFMA_PEAK_TFLOPS, 4),
                                      "whole_pass": {"ms": round(ms, 4), "tflops": round(tf, 2), "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                                                     "crops_per_s_device": round(nc / (ms * 1e-3), 1)},
                                      "all_kernels": {k: {"ms": round(v[0] / 3, 4), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else None}
                                                      for k, v in sorted(ragg.items(), key=lambda kv: -kv[1][0])}}
                del crops, labels, probs
            roof_rec["kernel"] = roof_rec["b65536"]["kernel"]
            roof_rec["achieved"] = roof_rec["b65536"]["achieved"]
            roof_rec["frac"] = roof_rec["b65536"]["frac"]
            extras["roofline_rec"] = roof_rec
            # EXTENSION (never the headline; the reference has no sequence recogniser): CTC greedy decode of BASELINE configs[2]'s literal
            # shape - 256 crops of 32 x 128 -> T = 32 columns, C = 63 classes (utils.rs:7 alphabet + blank), logits resident in HBM
            try:
                nc, tt, cc = 256, 32, 63
                g = torch.Generator(device="cpu").manual_seed(5)
                lg = torch.randn((nc, tt, cc), generator=g).to(x.device)
                lab = torch.empty((nc, tt), dtype=torch.int32, device=x.device)
                ln = torch.empty(nc, dtype=torch.int32, device=x.device)
                torch.cuda.synchronize()
                for _ in range(3):
                    rec.ctc_greedy_decode_device(lg.data_ptr(), nc, tt, cc, cc - 1, lab.data_ptr(), ln.data_ptr())
                t1 = time.perf_counter()
                it = 50
                for _ in range(it):
                    rec.ctc_greedy_decode_device(lg.data_ptr(), nc, tt, cc, cc - 1, lab.data_ptr(), ln.data_ptr())
                wall = time.perf_counter() - t1
                from oracle import ctc_oracle as CT   # (the checker, outside the timed region)
                wl, wn = CT.ctc_greedy_decode(lg.cpu().numpy(), cc - 1)
                ok = bool((lab.cpu().numpy() == wl).all() and (ln.cpu().numpy() == wn).all())
                extras["extension"] = {"ctc_decode_crops_per_s": round(nc * it / wall, 1), "crops": nc, "T": tt, "C": cc, "blank": cc - 1,
                                       "matches_oracle": ok, "bytes_per_call": nc * tt * (cc + 1) * 4,
                                       "note": "ocr_ctc_greedy_decode, blocking call per batch (launch + sync latency decides at this size); no reference counterpart"}
            except Exception as e:
                extras["extension"] = {"error": f"{type(e).__name__}: {e}"}
            rec.close()
        except Exception as e:  # side numbers never hide the headline
            extras["rec_error"] = f"{type(e).__name__}: {e}"

    total_images = n * world * a.steps
    exit_code = 0
    if rank == 0:
        line = {
            "metric": "images/sec (640x640 detect)", "value": round(total_images / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "arithmetic": ("f32 tensors and f32 accumulation everywhere; Winograd kernels and 1x1 convs on the f32 matrix instructions; the "
                           "other MFMA-bound convs, the stem and the head multiply on v_mfma_f32_32x32x16_bf16 from operands split into three "
                           "bf16 terms (six partial products per f32 product, dropped terms <= 2^-23 of it: the error of an f32 FMA chain, "
                           "profiles/r03_bf16x3_accuracy.txt; same parity bars as mfma=f32, whose rate is `f32_mfma_only`)") if a.dtype == "f32"
                          else "bf16 operands (activations and weights rounded to bf16), f32 accumulation: the opt-in precision of configs[4]",
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
        line.update(strict)
        if bf16:
            line["bf16"] = bf16
        if sweep:
            line["post_threads_sweep"] = sweep
        line.update(host)
        line.update(e2e)
        line.update(post)
        line.update(extras)
        if a.det_options:
            line["det_options"] = a.det_options
        line["post_threads"] = min(16, cores_per_rank)
        line["cores_per_rank"] = cores_per_rank
        line["rank_cpus"] = PINNED_CPUS   # N > 1: the CPUs this rank pinned itself to before any GPU call (nearest its GPU's NUMA node); None = unpinned
        if not a.no_cpu_baseline:   # every N: each line of a scaling sweep stands alone (the other ranks wait in the barrier below)
            try:
                line["cpu_baseline"] = cpu_baseline(det_w, rec_w, s, a.cpu_seconds)
            except Exception as e:
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        if dist is not None and try_c_abi and gathered is not None:
            line.update(c_abi_exchange(capi, dist, torch, polys, scores, gathered, world, rank, local, line))
        # the last ~1 500 characters of the line are what a stored record keeps: the numbers a reader needs, once more, at the very end
        cb = line.get("cpu_baseline") or {}
        b16 = line.get("bf16") or {}
        sw = (line.get("post_threads_sweep") or {}).get("2") or {}
        line["summary"] = {
            "ms_per_step": line["ms_per_step"], "images_per_s": line["value"], "n_gpus": world, "dtype": a.dtype,
            "roofline": {"kernel": roof.get("kernel"), "frac": roof.get("frac"), "achieved_tflops": roof.get("achieved"), "peak_tflops": roof.get("peak"),
                         "avg_launch_ms": roof.get("avg_launch_ms"), "avg_launch_ms_in_schedule": roof.get("avg_launch_ms_in_schedule"), "traffic_bytes": roof.get("traffic"), "mfma_busy": roof.get("mfma_busy")},
            "f32_mfma_only_ms": (line.get("f32_mfma_only") or {}).get("ms_per_step"),
            "bf16": {k2: b16.get(k2) for k2 in ("ms_per_step", "images_per_s", "e2e_pages_per_s")} if b16 else None,
            "e2e_pages_per_s": line.get("e2e_pages_per_s"),
            "detect_postprocess_pipelined_images_per_s": line.get("detect_postprocess_pipelined_images_per_s"),
            "postprocess_images_per_s": {"text": line.get("postprocess_images_per_s"), "dense": line.get("postprocess_dense_images_per_s")},
            "post_threads_2": {k2: v2.get("detect_postprocess_pipelined_images_per_s") for k2, v2 in sw.items() if isinstance(v2, dict) and "detect_postprocess_pipelined_images_per_s" in v2} if sw else None,
            "post_threads_16_device_chain": {k2: v2.get("detect_postprocess_pipelined_images_per_s") for k2, v2 in ((line.get("post_threads_sweep") or {}).get("16_device_chain") or {}).items()
                                             if isinstance(v2, dict) and "detect_postprocess_pipelined_images_per_s" in v2} or None,
            "extension": line.get("extension"),
            "rec_crops_per_s": {"b256": line.get("rec_crops_per_s_b256"), "b65536": line.get("rec_crops_per_s_b65536")},
            "cpu_baseline": {"images_per_s": cb.get("value"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "one_thread": (cb.get("one_thread") or {}).get("value"),
                             "rec_crops_per_s": (cb.get("recognition") or {}).get("value"),
                             "postprocess_images_per_s": (cb.get("postprocess") or {}).get("value")},
            "exchange_ok": line.get("exchange_ok"),
        }
        emit(line)
        if line.get("exchange_ok") is False and backend == "nccl":
            exit_code = 4   # the line is out, with the error in it; a failed exchange over RCCL must not read as a clean run
    elif dist is not None and try_c_abi and gathered is not None:
        c_abi_exchange(capi, dist, torch, polys, scores, gathered, world, rank, local, None)
    det.close()
    if dist is not None:
        try:   # no closing barrier: the launcher waits for every rank, and a rank that left through the deadline in
            dist.destroy_process_group()   # c_abi_exchange must not stall the others after the line is out
        except Exception:
            pass
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()

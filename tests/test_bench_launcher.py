"""`python bench.py --gpus N` as a bare command: the parent must start N ranks as child processes (before any
GPU call), relay rank 0's JSON line and fail when a rank fails.  CPU-only: --dry-run swaps the GPU work for a
gloo rendezvous + the result all-gather, the launcher code is the one the GPU run uses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bare_command_launches_its_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--batch", "7", "--steps", "4", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout   # ONE line, from rank 0: library banners (gloo, RCCL) go to stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1
    assert line["all_gather_results"]["images"] == 14   # both shards arrived, in rank order


def test_failing_rank_fails_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(OCR_BENCH_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_runs_as_a_rank_under_an_external_launcher():
    # the shape torch.distributed.run gives: RANK / WORLD_SIZE already set -> no second level of children
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_launcher_makes_no_gpu_call():
    """The parent must not import torch or the library before it forks: nothing that could initialise HIP."""
    src = open(BENCH).read()
    head = src[:src.index("def main():")]
    top_level = [ln for ln in head.splitlines() if ln.startswith(("import ", "from "))]
    assert not any("torch" in ln or "ocr_rs_amd" in ln or "numpy" in ln for ln in top_level), top_level


def test_world_8_rendezvous_and_core_shares():
    """The shape of the driver's scaling run (--gpus 8): eight ranks rendezvous, every shard arrives in rank order, and
    each rank is told its share of the host cores (what sizes its post-processing pool)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run", "--batch", "3"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["all_gather_results"]["images"] == 24
    sys.path.insert(0, os.path.dirname(BENCH))
    import bench
    assert line["cores_per_rank"] == max(1, bench.host_cores() // 8)
    # ... and has pinned itself to it: eight pairwise disjoint masks of that size (when the host has at least eight CPUs)
    if len(os.sched_getaffinity(0)) >= 8:
        masks = line["rank_cpus"]
        assert line["pinned"] and len(masks) == 8
        assert all(len(m) == line["cores_per_rank"] for m in masks), masks
        assert len(set(c for m in masks for c in m)) == sum(len(m) for m in masks), masks


def test_affinity_plan_follows_the_gpus_numa_nodes(tmp_path):
    """plan_affinity / gpu_local_cpus on a made-up two-socket host (sysfs tree in a temp directory): 8 GPUs, four per NUMA node,
    128 CPUs - each rank gets 16 CPUs of ITS GPU's node, all masks disjoint; without topology: contiguous slices."""
    sys.path.insert(0, os.path.dirname(BENCH))
    import bench
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for i in range(2):   # CPU nodes first, as KFD lists them
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for g in range(8):
        d = nodes / str(2 + g)
        d.mkdir(parents=True)
        bus = 0x10 + 0x10 * g
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id {bus << 8}\n")
        pci = tmp_path / "bus" / "pci" / "devices" / f"0000:{bus:02x}:00.0"
        pci.mkdir(parents=True)
        (pci / "local_cpulist").write_text("0-31,64-95\n" if g < 4 else "32-63,96-127\n")
    old = {k: os.environ.pop(k, None) for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES")}
    try:
        gpus = bench.gpu_local_cpus(str(tmp_path))
    finally:
        os.environ.update({k: v for k, v in old.items() if v is not None})
    assert len(gpus) == 8 and gpus[0] == set(range(0, 32)) | set(range(64, 96)) and gpus[7] == set(range(32, 64)) | set(range(96, 128))
    plan = bench.plan_affinity(8, range(128), 16, gpus)
    assert all(len(m) == 16 for m in plan)
    assert len(set(c for m in plan for c in m)) == 128
    assert all(set(plan[r]) <= gpus[r] for r in range(8))
    assert plan[0] == list(range(16)) and plan[4] == list(range(32, 48))
    # unknown topology / a restricted mask: contiguous slices of what is allowed
    plan = bench.plan_affinity(8, range(8, 40), 4, None)
    assert plan == [list(range(8 + 4 * r, 12 + 4 * r)) for r in range(8)]
    # a GPU whose node has too few free CPUs borrows from the rest, still disjoint
    plan = bench.plan_affinity(2, range(8), 4, [{0, 1}, {0, 1}])
    assert plan[0][:2] == [0, 1] and len(set(plan[0]) | set(plan[1])) == 8
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_failed_exchange_is_never_a_clean_exit():
    """A stalled or failed C-ABI exchange emits the line WITH the error and exits non-zero (rank 0 and with it the launcher)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(OCR_BENCH_FAIL_EXCHANGE="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["exchange_ok"] is False


def test_committed_counter_extract_belongs_to_the_committed_kernels():
    """profiles/pmc.json (HBM traffic, MFMA counters, in-schedule launch averages that bench.py's roofline block cites) is stamped with a
    hash of ocr-rs_amd/csrc taken on the GPU box: the committed extract must be the one of the committed kernel sources, or the line
    says `traffic_stale: true`."""
    sys.path.insert(0, os.path.dirname(BENCH))
    import bench
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))
    assert j["f32"]["csrc_sha"] == j["bf16"]["csrc_sha"] == bench.csrc_hash()
    k, src, stale = bench.pmc_extract("f32", 32, 640)
    assert stale is False and "winograd43_fused<c64>" in k
    assert bench.in_schedule_ms("winograd43_fused<c64>") > 0.15

"""bench.py's N > 1 path on the GPU box: the bare command starts two ranks itself, both run the detector, post-process
their shard and all-gather the polygon lists; rank 0 prints ONE line.  A 1-GPU box cannot host two RCCL ranks (RCCL
refuses two ranks on one device), so the collectives go through gloo here (OCR_BENCH_BACKEND=gloo: the rehearsal switch
documented in bench.py) and OCR_BENCH_C_ABI=1 drives the library's own communicator into exactly that refusal - the
error path that must never take the line down."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_print_one_line():
    env = dict(os.environ, OCR_BENCH_BACKEND="gloo", OCR_BENCH_C_ABI="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 64
    assert line["all_gather_results"]["images"] == 64            # both shards arrived
    assert line["all_gather_results_ms"] > 0 and line["detect_postprocess_gather_images_per_s"] > 0
    assert line["rccl_ranks"] == 0 and line["collective_backend"] == "gloo"
    # the C-ABI exchange either ran (two devices) or reported why not - it never took the line down
    assert ("all_gather_results_c_abi_ms" in line) != ("all_gather_results_c_abi_error" in line)

"""world_size-2 tests of the sharding / result all-gather path on gloo (CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import parallel as P


def test_shard_range_covers_batch():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip_ragged():
    polys = [[[(1, 2), (3, 4), (5, 6), (7, 8)]], [], [[(9, 9)] * 5, [(0, 4294967295)] * 4]]
    scores = [[0.9819034852546917], [], [0.7, float("nan")]]
    p2, s2 = P.unpack_results(P.pack_results(polys, scores))
    assert p2 == polys
    assert s2[0] == scores[0] and s2[1] == [] and s2[2][0] == 0.7 and np.isnan(s2[2][1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_results(img):
    # deterministic, ragged per-image results (image i has i % 3 polygons)
    polys = [[(img * 10 + k, img + v) for v in range(4 + k)] for k in range(img % 3)]
    return polys, [0.7 + 0.01 * k + 0.001 * img for k in range(img % 3)]


def _worker(rank, world, port, n_images, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = P.shard_range(n_images, world, rank)
    polys, scores = zip(*[_fake_results(i) for i in range(lo, hi)]) if hi > lo else ((), ())
    all_p, all_s = P.all_gather_results(list(polys), list(scores), torch.device("cpu"))
    labels = torch.arange(lo, hi, dtype=torch.int32)
    counts = [P.shard_range(n_images, world, r)[1] - P.shard_range(n_images, world, r)[0] for r in range(world)]
    all_l = P.all_gather_labels(labels, counts)
    exp_p = [_fake_results(i)[0] for i in range(n_images)]
    exp_s = [_fake_results(i)[1] for i in range(n_images)]
    ok = all_p == exp_p and all_s == exp_s and all_l.tolist() == list(range(n_images))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_results_world2():
    world, n_images = 2, 7            # ragged: 4 + 3 images, variable polygons per image
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}

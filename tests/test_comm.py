"""The result all-gather behind the C ABI (ocr_comm_*, comm.hip): wire format and assembly on the CPU through the
test hook, the RCCL transport itself on the GPU with the one rank a 1-GPU box can host."""
import numpy as np
import pytest

import ocr_rs_amd  # noqa: F401
from ocr_rs_amd import capi
from ocr_rs_amd import parallel as P


def _shard(rank, n_images):
    polys = [[[(rank * 1000 + i * 10 + k, v + i) for v in range(4 + k)] for k in range((i + rank) % 3)] for i in range(n_images)]
    scores = [[0.7 + 0.01 * k + 0.001 * i + 0.0001 * rank for k in range((i + rank) % 3)] for i in range(n_images)]
    return polys, scores


@pytest.mark.parametrize("sizes", [[4, 3], [1, 0, 5, 2], [0, 0], [7]])
def test_assembly_is_rank_order_concatenation(sizes):
    shards = [_shard(r, n) for r, n in enumerate(sizes)]
    polys, scores = capi.comm_assemble(shards)
    assert polys == [p for s in shards for p in s[0]]
    assert scores == [x for s in shards for x in s[1]]
    # ... and equals what the torch.distributed path of ocr-rs_amd/parallel.py reconstructs from its own packing
    ref = [P.unpack_results(P.pack_results(*s)) for s in shards]
    assert polys == [p for r in ref for p in r[0]] and scores == [x for r in ref for x in r[1]]


def test_extreme_values_survive_the_wire():
    polys = [[[(0, 4294967295), (4294967295, 0), (1, 2), (3, 4)]], []]
    scores = [[float("nan")], []]
    p2, s2 = capi.comm_assemble([(polys, scores), ([], [])])
    assert p2 == polys and np.isnan(s2[0][0]) and s2[1] == []


@pytest.mark.gpu
def test_rccl_transport_world_size_one():
    """ncclCommInitRank + the two ncclAllGather calls on the card (a 1-GPU box hosts one rank; the N-rank run is the
    driver's bench.py --gpus N, which reports all_gather_results_c_abi_ms)."""
    assert capi.Comm.rccl_version() > 20000
    comm = capi.Comm(capi.Comm.unique_id(), 1, 0, 0)
    polys, scores = _shard(0, 9)
    for _ in range(3):                       # buffers are reused and grown across calls
        p2, s2 = comm.all_gather_polygons(polys, scores)
        assert p2 == polys and s2 == scores
    big = _shard(0, 400)
    assert comm.all_gather_polygons(*big) == (big[0], big[1])
    assert comm.all_gather_polygons([], []) == ([], [])
    labels = np.arange(1001, dtype=np.int32) % 62
    allv, counts = comm.all_gather_labels(labels, 2048)
    assert np.array_equal(allv, labels) and counts.tolist() == [1001]
    with pytest.raises(capi.OcrError):
        comm.all_gather_labels(labels, 10)    # capacity too small: an error, not an overrun
    comm.close()


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_two_ranks_on_one_device_fail_cleanly_from_two_threads():
    """The error path of the communicator a multi-threaded host can hit: two ranks (two threads of ONE process, the
    one-process-many-GPUs form the header allows) both name device 0.  RCCL refuses a duplicate device; both ocr_comm_create
    calls must come back with an error message - no hang, no crash - and the library must stay usable afterwards."""
    import threading
    uid = capi.Comm.unique_id()
    out = [None, None]

    def rank(r):
        try:
            c = capi.Comm(uid, 2, r, 0)
            c.close()
            out[r] = "created"
        except capi.OcrError as e:
            out[r] = e

    ts = [threading.Thread(target=rank, args=(r,), daemon=True) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(150)
    assert not any(t.is_alive() for t in ts), "ocr_comm_create did not return"
    assert all(isinstance(o, capi.OcrError) for o in out), out
    assert all("ncclCommInitRank" in str(o) or "RCCL" in str(o) or "nccl" in str(o).lower() for o in out), out
    comm = capi.Comm(capi.Comm.unique_id(), 1, 0, 0)          # a fresh single-rank communicator still works
    polys, scores = _shard(0, 3)
    assert comm.all_gather_polygons(polys, scores) == (polys, scores)
    comm.close()

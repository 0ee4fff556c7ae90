"""Multi-GPU layer of the hot path: embarrassingly parallel frame/crop sharding, one
process per GPU, and ONE exchange step - an all-gather of the variable-length result
blocks.  The reference has no distributed code at all; this is new.

Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" for the
CPU test-suite.  The payload is KB-scale, so the collective is latency-bound: a
fixed-size header all-gather (polygon and vertex counts) followed by one payload
all-gather padded to the largest rank.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(polys: Sequence[Sequence[Sequence[Tuple[int, int]]]], scores: Sequence[Sequence[float]]):
    """PolygonScores of the local images -> (header int64[2], payload int64[]) flat CSR."""
    img_counts = [len(p) for p in polys]
    poly_lens = [len(pg) for p in polys for pg in p]
    xy = [c for p in polys for pg in p for v in pg for c in v]
    sc = np.asarray([s for ss in scores for s in ss], dtype=np.float64).view(np.int64)
    payload = np.concatenate([
        np.asarray([len(img_counts), len(poly_lens), len(xy)], dtype=np.int64),
        np.asarray(img_counts, dtype=np.int64), np.asarray(poly_lens, dtype=np.int64),
        np.asarray(xy, dtype=np.int64), sc.astype(np.int64)])
    return payload


def unpack_results(payload: np.ndarray):
    n_img, n_poly, n_xy = (int(v) for v in payload[:3])
    pos = 3
    img_counts = payload[pos:pos + n_img]; pos += n_img
    poly_lens = payload[pos:pos + n_poly]; pos += n_poly
    xy = payload[pos:pos + n_xy]; pos += n_xy
    sc = payload[pos:pos + n_poly].astype(np.int64).view(np.float64)
    polys, scores = [], []
    pi = vi = 0
    for c in img_counts:
        ip, isc = [], []
        for _ in range(int(c)):
            L = int(poly_lens[pi])
            ip.append([(int(xy[2 * (vi + k)]), int(xy[2 * (vi + k) + 1])) for k in range(L)])
            isc.append(float(sc[pi]))
            vi += L
            pi += 1
        polys.append(ip)
        scores.append(isc)
    return polys, scores


def all_gather_results(polys, scores, device: torch.device):
    """Every rank receives the PolygonScores of the whole batch, in rank (= frame) order."""
    world = dist.get_world_size()
    payload = torch.from_numpy(pack_results(polys, scores)).to(device)
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)                                  # header: payload length per rank
    mx = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(mx, dtype=torch.int64, device=device)
    padded[:payload.numel()] = payload
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)                                  # payload, padded to the largest rank
    all_p, all_s = [], []
    for r in range(world):
        p, s = unpack_results(out[r][:int(sizes[r].item())].cpu().numpy())
        all_p += p
        all_s += s
    return all_p, all_s


def all_gather_labels(labels: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """Recognition labels (int32[n_local]) of every rank, concatenated in rank order."""
    world = dist.get_world_size()
    mx = max(counts)
    padded = torch.zeros(mx, dtype=labels.dtype, device=labels.device)
    padded[:labels.numel()] = labels
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([out[r][:counts[r]] for r in range(world)])

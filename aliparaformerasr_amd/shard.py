"""Utterance-level data parallelism for the offline path.

Utterances are independent (the batch is dim 0 of every tensor the reference builds,
AliParaformerAsr/OfflineProjOfParaformer.cs:49), so N GPUs = N shards of the utterance list,
one process per GPU, NO data-path collective.  The only collectives are
  * the one-off broadcast of the PFW weight image (rank 0 -> all), and
  * the gather of fixed-shape int32 hypotheses, restored to the caller's original order.
Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous blocks of ceil(n/world) (B_g = ceil(B/G))."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi


def shard_list(items, world: int, rank: int):
    lo, hi = shard_bounds(len(items), world, rank)
    return items[lo:hi]


def broadcast_bytes(blob, dist, device, src: int = 0):
    """Broadcast a bytes-like object from `src`; returns a uint8 torch tensor on `device`
    (on the GPU box the engine adopts that device image in place: weights_device)."""
    import torch
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, src)
    return t


def gather_hypotheses(ids_local: np.ndarray, n_total: int, lcap: int, dist, device):
    """ids_local [B_local, L] int64 -> [n_total, lcap] int32 on every rank (right-padded with -1),
    rows in the original utterance order."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    buf = np.full((per, lcap), -1, np.int32)
    b, L = ids_local.shape if ids_local.size else (ids_local.shape[0], 0)
    if L > lcap:
        raise ValueError("hypothesis longer than lcap")
    buf[:b, :L] = ids_local
    mine = torch.from_numpy(buf).to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    full = torch.cat(out, dim=0).cpu().numpy()
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        rows.append(full[r * per: r * per + (hi - lo)])
    return np.concatenate(rows, axis=0)


def gather_hypotheses_device(mine, n_total: int, dist):
    """`mine` [per, lcap] integer tensor ALREADY on the collective's device (the engine wrote it there,
    pf_fetch_ids_device; rows past this rank's shard and columns past L hold -1) -> [n_total, lcap] on every rank in
    the original utterance order, still on the device: one all_gather (RCCL over xGMI on the GPU box), no host copy."""
    import torch
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    assert mine.shape[0] == per
    full = torch.empty((world * per,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(full, mine.contiguous())
    if world * per == n_total:
        return full
    keep = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        keep.append(full[r * per: r * per + (hi - lo)])
    return torch.cat(keep, dim=0)

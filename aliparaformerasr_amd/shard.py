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


class StepPipeline:
    """K steps of a job over E engines on one device, several steps in flight (bench.py --in-flight E).

    Global step i runs on engine i % E: `step_fn(e, parity)` is called by engine e's own host thread (a step blocks its
    thread at the decoder-length read-back, so one thread cannot keep two engines busy); an engine's steps run in order.
    `after_fn(e, parity)` — the collective that gathers the step's hypotheses — is called by the thread that called
    `run`, in GLOBAL step order 0, 1, 2, ..: every rank therefore enters the collectives in the same order whatever the
    pace of its engines, with one communicator.  `parity` = (i // E) & 1 names which of the engine's two output buffers
    the step fills; an engine runs at most two steps ahead of its gathers, so a buffer is never refilled before it has
    been gathered.  An exception in a step is re-raised by `run` after the threads have been joined."""

    def __init__(self, n_engines, step_fn, after_fn=None):
        self.E = int(n_engines)
        self.step_fn = step_fn
        self.after_fn = after_fn

    def run(self, first, count):
        import queue
        import threading
        E = self.E
        if E == 1 or count <= 1:
            for i in range(first, first + count):
                self.step_fn(i % E, (i // E) & 1)
                if self.after_fn:
                    self.after_fn(i % E, (i // E) & 1)
            return
        done = [queue.Queue() for _ in range(E)]
        room = [threading.Semaphore(2) for _ in range(E)]
        errs = []
        stop = threading.Event()

        def worker(e):
            try:
                for i in range(first, first + count):
                    if i % E != e:
                        continue
                    room[e].acquire()
                    if stop.is_set():
                        return
                    self.step_fn(e, (i // E) & 1)
                    done[e].put(i)
            except BaseException as ex:                      # noqa: BLE001 — relayed to the caller
                errs.append(ex)
                done[e].put(-1)

        th = [threading.Thread(target=worker, args=(e,)) for e in range(E)]
        for t in th:
            t.start()
        try:
            for i in range(first, first + count):
                got = done[i % E].get()
                if got < 0:
                    break
                assert got == i, (got, i)
                if self.after_fn:
                    self.after_fn(i % E, (i // E) & 1)
                room[i % E].release()
        finally:
            stop.set()                                       # a failed run must not leave a worker waiting for room
            for r in room:
                r.release()
            for t in th:
                t.join()
        if errs:
            raise errs[0]

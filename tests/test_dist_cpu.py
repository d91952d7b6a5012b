"""CPU, world_size 2 (gloo): the N>1 plumbing of bench.py / shard.py — shard bounds, weight-image
broadcast, hypothesis gather with order restore."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aliparaformerasr_amd import shard as sh
from aliparaformerasr_amd import weights as W


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 32, 1024, 1025):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = sh.shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n
                got += list(range(lo, hi))
            assert got == list(range(n))
    assert sh.shard_bounds(1024, 8, 3) == (384, 512)          # config 4: 128 utterances per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # 1. weight image: rank 0 packs, everyone receives identical bytes and parses the same header
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=32)
    blob = W.pack_pfw(cfg, W.synth_weights(cfg, 3)) if rank == 0 else b""
    t = sh.broadcast_bytes(blob, dist, dev)
    cfg2, w2 = W.load_pfw(bytes(t.numpy().tobytes()))
    # 2. each rank "recognises" its shard: hypothesis of utterance u = [u, u+1, ...] of length 3 + u % 5
    lo, hi = sh.shard_bounds(n_total, world, rank)
    L = max([3 + u % 5 for u in range(lo, hi)] + [0])
    ids = np.full((hi - lo, L), -1, np.int64)
    for i, u in enumerate(range(lo, hi)):
        ids[i, : 3 + u % 5] = np.arange(u, u + 3 + u % 5)
    full = sh.gather_hypotheses(ids, n_total, 16, dist, dev)
    # 3. the tensor form bench.py uses on the GPU box (ids already on the collective's device, one all_gather_into_tensor)
    per = (n_total + world - 1) // world
    mine = torch.full((per, 16), -1, dtype=torch.int64)
    mine[: hi - lo, :L] = torch.from_numpy(ids)
    full_t = sh.gather_hypotheses_device(mine, n_total, dist).numpy()
    assert full_t.shape == full.shape and (full_t == full).all()
    q.put((rank, cfg2 == cfg, float(w2["decoder.output.weight"].sum()), full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8])
def test_broadcast_and_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda o: o[0])
    assert all(o[1] for o in outs)
    assert outs[0][2] == outs[1][2]
    for _, _, _, full in outs:
        assert full.shape == (n_total, 16)
        for u in range(n_total):
            n = 3 + u % 5
            np.testing.assert_array_equal(full[u, :n], np.arange(u, u + n))
            assert (full[u, n:] == -1).all()

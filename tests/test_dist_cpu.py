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


def test_broadcast_and_gather_world4_unequal_shards():
    """1025 utterances over 4 ranks: shards of 257 / 257 / 257 / 254 (ceil(B/G) blocks, the last one short) — the padded
    rows of the short shard must be dropped and the caller's order restored, in both gather forms."""
    n_total, world = 1025, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [sh.shard_bounds(n_total, world, r) for r in range(world)] == [(0, 257), (257, 514), (514, 771), (771, 1025)]
    assert all(o[1] for o in outs) and len({o[2] for o in outs}) == 1
    for _, _, _, full in outs:
        assert full.shape == (n_total, 16)
        for u in range(n_total):
            n = 3 + u % 5
            np.testing.assert_array_equal(full[u, :n], np.arange(u, u + n))
            assert (full[u, n:] == -1).all()


def _pipeline_worker(rank, world, port, E, steps, q):
    """bench.py --gpus N --in-flight E in miniature: E fake engines per rank whose steps take random, rank-dependent times;
    the gathers must still pair step i of every rank (one communicator, collectives in global step order)."""
    import random
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = random.Random(100 + rank)
    bufs = [[torch.full((2, 4), -1, dtype=torch.int64) for _ in range(2)] for _ in range(E)]
    nxt = [e for e in range(E)]                        # the global step engine e runs next
    seen = []

    def step(e, par):
        time.sleep(rnd.uniform(0.0, 0.004) * (1 + 3 * ((rank + e) % 2)))
        bufs[e][par].fill_(rank * 1000 + nxt[e])
        nxt[e] += E

    def after(e, par):
        full = sh.gather_hypotheses_device(bufs[e][par], 2 * world, dist)
        seen.append(full[:, 0].tolist())

    pipe = sh.StepPipeline(E, step, after)
    pipe.run(0, 3)                                      # "warm-up", then the timed region split as bench.py splits it
    for e in range(E):                                  # (bench.py restarts its counters the same way: steps are global)
        nxt[e] = e
    seen.clear()
    pipe.run(0, 1)
    pipe.run(1, steps - 1)
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("E", [1, 2, 3])
def test_step_pipeline_keeps_collectives_in_global_step_order_world2(E):
    world, steps = 2, 11
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, E, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert len(out[r]) == steps
        for i, row in enumerate(out[r]):                # step i of EVERY rank, in rank order, two rows per rank
            assert row == [0 * 1000 + i, 0 * 1000 + i, 1 * 1000 + i, 1 * 1000 + i], (E, r, i, row)


def test_step_pipeline_relays_a_failing_step():
    calls = []

    def step(e, par):
        calls.append(e)
        if len(calls) == 5:
            raise RuntimeError("engine failed")

    with pytest.raises(RuntimeError, match="engine failed"):
        sh.StepPipeline(2, step).run(0, 40)
    assert len(calls) < 40

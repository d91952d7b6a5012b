"""GPU: pf_group — several engines inside one process, through the C ABI.

The GPU box has ONE device, so the group is exercised as (a) devices = [0]: a one-rank RCCL communicator, the
broadcast / all-gather code path with no traffic, and (b) devices = [0, 0] / [0, 0, 0]: two / three engines, worker
threads, shards, the batch-wide padding and the batch-wide decoder length, hypotheses merged in the caller's order
(no communicator: RCCL refuses a repeated device).  The N-distinct-GPU communicator itself is only reachable on a
multi-GPU node and is unmeasured here.
"""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=300, timestamp_head=True)
    w = W.synth_weights(cfg, 21)
    return cfg, W.pack_pfw(cfg, w), W.synth_cmvn()


def _single(model, audio, **kw):
    from aliparaformerasr_amd.engine import Engine
    cfg, blob, cmvn = model
    e = Engine(weights=blob, cmvn=cmvn, device=0)
    r = e.recognize(audio, **kw)
    e.close()
    return r


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_equals_single_engine(model, devices):
    from aliparaformerasr_amd.engine import EngineGroup
    cfg, blob, cmvn = model
    # ragged lengths, B not a multiple of the group size: the LAST shard holds the longest utterance, so every other
    # shard must be padded beyond its own maximum (PadHelper.cs:25) and decode more positions than its own fires
    audio = [W.synth_audio(n, u) for u, n in enumerate((32000, 40000, 24000, 36000, 30000, 28000, 64000))]
    ref = _single(model, audio, want_logits=True)
    g = EngineGroup(devices, weights=blob, cmvn=cmvn)
    assert g.size == len(devices)
    assert g.uses_rccl == (len(devices) == 1)
    got = g.recognize(audio, want_logits=True)
    assert got.L == ref.L
    np.testing.assert_array_equal(got.token_num, ref.token_num)
    np.testing.assert_array_equal(got.token_ids, ref.token_ids)
    # utterances are independent given the same padding: the log-probs agree to reduction-order noise
    np.testing.assert_allclose(got.logits, ref.logits, atol=2e-3)
    np.testing.assert_allclose(got.cif_peak, ref.cif_peak, atol=2e-3)
    # ids-only call, fewer utterances than engines, then an empty list
    r2 = g.recognize(audio[:1])
    np.testing.assert_array_equal(r2.token_ids, _single(model, audio[:1]).token_ids)
    assert g.recognize([]).token_ids.shape[0] == 0
    g.close()
    g.close()


def test_group_failure_on_one_shard_does_not_hang(model):
    """A shard whose forward throws (here: every utterance of the batch shorter than one LFR frame => T = 0) must
    release the other workers from the decoder-length rendezvous and surface as an error, not as a dead-lock."""
    from aliparaformerasr_amd._native import PfError
    from aliparaformerasr_amd.engine import EngineGroup
    cfg, blob, cmvn = model
    g = EngineGroup([0, 0], weights=blob, cmvn=cmvn)
    with pytest.raises(PfError):
        g.recognize([np.zeros(100, np.float32), np.zeros(200, np.float32)])
    ok = g.recognize([W.synth_audio(32000, 1), W.synth_audio(32000, 2)])       # the group is still usable
    assert ok.token_ids.shape[0] == 2
    g.close()


def test_group_sensevoice_and_seaco(sv_embed):
    from aliparaformerasr_amd.engine import Engine, EngineGroup
    from oracle import glue
    cmvn = W.synth_cmvn()
    cfg = W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=400, use_itn=True)
    w = W.synth_weights(cfg, 5)
    w["embed.weight"] = sv_embed
    blob = W.pack_pfw(cfg, w)
    audio = [W.synth_audio(n, 40 + u) for u, n in enumerate((32000, 48000, 40000))]
    e = Engine(weights=blob, cmvn=cmvn, device=0)
    ref = e.recognize(audio)
    e.close()
    g = EngineGroup([0, 0], weights=blob, cmvn=cmvn)
    got = g.recognize(audio)
    np.testing.assert_array_equal(got.token_ids, ref.token_ids)
    g.close()
    cfg = W.seaco_paraformer_config(enc_layers=2, dec_layers=2, seaco_layers=1, vocab=300, seaco_nobias=290)
    blob = W.pack_pfw(cfg, W.synth_weights(cfg, 6))
    hw = np.asarray(glue.pad_list([[11, 12], [100, 200, 30], [1]]), np.int32)
    e = Engine(weights=blob, cmvn=cmvn, device=0)
    ref = e.recognize(audio, hotwords=hw)
    e.close()
    g = EngineGroup([0, 0], weights=blob, cmvn=cmvn)
    got = g.recognize(audio, hotwords=hw)
    np.testing.assert_array_equal(got.token_ids, ref.token_ids)
    np.testing.assert_allclose(got.cif_peak, ref.cif_peak, atol=2e-3)
    g.close()


def test_bench_rccl_path_with_a_one_rank_communicator():
    """bench.py's multi-GPU line as far as a 1-GPU box can execute it: one rank under torch.distributed.run with
    PF_BENCH_DIST=1 — RCCL process group, weight image broadcast into device memory and adopted in place, ids written
    into a device tensor by pf_fetch_ids_device, all_gather_into_tensor, gathered rows == the rank's own ids (asserted
    inside bench.py), ids checked against the fp32 oracle's golden file."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PF_BENCH_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "7", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    # two steps in flight: both engines' steps were gathered by the main thread in global step order (shard.StepPipeline)
    assert d["config"]["steps_in_flight"] == 2 and d["steps"] == 7
    assert d["ids_vs_fp32_oracle"] and d["ids_vs_fp32_oracle"]["ok"]

"""CPU: the device-independent half of pf_group — shard plan, rendez-vous, fixed-shape gather blocks, failure release
and the merge in the caller's order (csrc/shards.cpp) — driven through `pf_host_group_sim`, which runs the real
ShardRunner with arithmetic stand-ins for the devices (utterance u decodes to ids[u][l] = u * 100000 + l).

Covers what a 1-GPU box cannot: G = 2, 3, 8 (and 64) shards with ragged B, empty shards ((G-1) * ceil(B/G) >= B, e.g.
G = 8, B = 9 or B < G), models without a decoder-length rendez-vous (SenseVoice) — where every rank must still present
the same all-gather count — and a shard failing before / after the decoder-length rendez-vous or while preparing the
gather: an error, never a hang (pytest-timeout guards the "never").
"""
import ctypes as C

import numpy as np
import pytest

from aliparaformerasr_amd import _native as N


def _sim(G, fire, has_cif=True, fixed_L=0, collective=True, fail_shard=-1, fail_stage=0):
    lib = N.load()
    fire = np.ascontiguousarray(fire, np.int32)
    B = fire.shape[0]
    cap = max(int(fire.max()) if B else 0, fixed_L, 1)
    ids = np.full((max(B, 1), cap), -7, np.int64)
    tn = np.full(max(B, 1), -7, np.int32)
    L = C.c_int32(-1)
    rc = lib.pf_host_group_sim(G, B, fire.ctypes.data_as(C.POINTER(C.c_int32)), 1 if has_cif else 0, fixed_L,
                               1 if collective else 0, fail_shard, fail_stage,
                               ids.ctypes.data_as(C.POINTER(C.c_int64)), cap, tn.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(L))
    return rc, L.value, ids[:B], tn[:B]


def _expect(fire, L):
    B = len(fire)
    return np.arange(B, dtype=np.int64)[:, None] * 100000 + np.arange(L, dtype=np.int64)[None, :]


@pytest.mark.timeout(60)
@pytest.mark.parametrize("collective", [True, False])
@pytest.mark.parametrize("G,B", [(1, 5), (2, 7), (3, 7), (8, 9), (8, 3), (8, 64), (8, 1024), (5, 1), (64, 100), (3, 0)])
def test_cif_models_merge_in_caller_order(G, B, collective):
    rng = np.random.default_rng(G * 1000 + B)
    fire = rng.integers(1, 40, size=B).astype(np.int32)
    rc, L, ids, tn = _sim(G, fire, collective=collective)
    assert rc == 0, N.load().pf_last_error()
    if B == 0:
        assert L == 0
        return
    assert L == int(fire.max())                       # the batch-wide decoder length, on every shard
    np.testing.assert_array_equal(ids[:, :L], _expect(fire, L))
    np.testing.assert_array_equal(tn, fire)


@pytest.mark.timeout(60)
@pytest.mark.parametrize("G,B", [(2, 3), (8, 9), (8, 3), (3, 64), (8, 1)])
def test_models_without_a_length_rendezvous_share_one_gather_count(G, B):
    """SenseVoice: L = T + 4 on every non-empty shard, nothing on an empty one — the all-gather count must still be the
    same on every rank (the stand-in collective refuses different counts; RCCL would hang or corrupt)."""
    fire = np.zeros(B, np.int32)
    rc, L, ids, tn = _sim(G, fire, has_cif=False, fixed_L=170)
    assert rc == 0, N.load().pf_last_error()
    assert L == 170
    np.testing.assert_array_equal(ids[:, :L], _expect(fire, L))
    np.testing.assert_array_equal(tn, np.full(B, 170, np.int32))


@pytest.mark.timeout(60)
@pytest.mark.parametrize("stage", [0, 1, 2])
@pytest.mark.parametrize("G,B,bad", [(2, 4, 0), (2, 4, 1), (3, 7, 1), (8, 9, 4), (8, 64, 7)])
def test_a_failing_shard_is_an_error_not_a_hang(G, B, bad, stage):
    fire = np.arange(1, B + 1, dtype=np.int32)
    rc, _, _, _ = _sim(G, fire, fail_shard=bad, fail_stage=stage)
    assert rc == N.PF_ERR_DEVICE                    # the root cause, not the "another device failed" echo
    msg = N.load().pf_last_error().decode()
    assert "simulated failure" in msg, msg
    # ... and the same call without the fault works afterwards (fresh runner per call in the stand-in; the real
    # group re-arms its barriers at the start of every call: tests/test_gpu_group.py)
    rc, L, ids, tn = _sim(G, fire)
    assert rc == 0 and L == B


@pytest.mark.timeout(60)
def test_failure_on_an_empty_shard_index_is_ignored_when_it_does_no_work():
    """fail_shard names a shard that owns no utterance (G = 8, B = 3 -> shards 3..7 are empty and are never run)."""
    fire = np.asarray([5, 2, 9], np.int32)
    rc, L, ids, tn = _sim(8, fire, fail_shard=6, fail_stage=1)
    assert rc == 0 and L == 9
    np.testing.assert_array_equal(ids[:, :L], _expect(fire, L))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("san", ["thread", "address"])
def test_shard_runner_under_sanitizers(san, tmp_path):
    """csrc/shards.cpp (worker threads, three rendez-vous, failure release, re-armed barriers) built with
    -fsanitize=thread / address around the stand-in backend and driven a few thousand times: no report, no crash."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / ("shards_" + san))
    cmd = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-g", "-O1", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-std=c++17",
           "-I" + os.path.join(root, "aliparaformerasr_amd", "csrc"), os.path.join(root, "tests", "native", "shards_sanitize.cpp"),
           os.path.join(root, "aliparaformerasr_amd", "csrc", "shards.cpp"), "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + b.stderr[-300:])
    r = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("ok ")

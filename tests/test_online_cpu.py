"""CPU: the host side of the streaming path (C++ in libparaformer_hip.so, no device needed) against the oracle
restatement of the reference's managed glue (oracle/online.py, citing OnlineStream.cs / OnlineWavFrontend.cs /
OnlineModel.cs / OnlineRecognizer.cs), plus hand-derived known answers."""
import ctypes as C

import numpy as np
import pytest

from aliparaformerasr_amd import _native as N
from oracle import online as oo


@pytest.fixture(scope="module")
def lib():
    return N.load()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_online_lfr_counts_and_values(lib):
    # OnlineWavFrontend.cs:63-80: t_lfr = t/6 - 1 when t % 6 < 1, else t/6; window i = frames 6i .. 6i+6, no left context
    for t, expect in ((61, 10), (60, 9), (7, 1), (6, 0), (5, 0), (13, 2), (12, 1)):
        fb = np.repeat(np.arange(1, t + 1, dtype=np.float32)[:, None], 80, axis=1)
        n = C.c_int32()
        out = np.zeros((max(expect, 1), 560), np.float32)
        N.check(lib.pf_host_online_lfr(_fp(fb), t, 7, 6, _fp(out), out.size, n))
        assert n.value == expect == oo.apply_lfr(fb).shape[0]
        if expect:
            np.testing.assert_array_equal(out[:expect], oo.apply_lfr(fb))
            assert out[0, 0] == 1 and out[0, 559] == 7              # frames 1..7 — NOT [0,0,0,1,2,3,4] as offline
            if expect > 1:
                assert out[1, 0] == 7 and out[1, 559] == 13
    # t = 0: t % 6 = 0 < 1 -> t_lfr = -1 -> the C# allocates a negative array: surfaces as a failure
    n = C.c_int32()
    assert lib.pf_host_online_lfr(None, 0, 7, 6, None, 0, n) == N.PF_ERR_RECOGNITION


def test_online_position_encoding_uses_i_plus_one(lib):
    rng = np.random.default_rng(0)
    for start in (0, 10, 250):
        x = rng.standard_normal((10, 560)).astype(np.float32)
        got = x.copy()
        N.check(lib.pf_host_online_posenc(_fp(got), 10, 560, start))
        ref = oo.position_encode(x, start)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)     # libm vs numpy sin/cos of the same float argument
        # known answer: row 0 at start 0 is position 1, first sin entry = sin(exp(-1 * ln(1e4)/279))
        if start == 0:
            inv0 = np.exp(-np.log(10000.0) / 279.0)
            assert abs((got[0, 0] - x[0, 0]) - np.sin(inv0)) < 1e-6
            assert abs((got[0, 280] - x[0, 280]) - np.cos(inv0)) < 1e-6


def test_online_dynamic_mask(lib):
    a = np.arange(1, 21, dtype=np.float32)
    got = a.copy()
    N.check(lib.pf_host_online_dynamic_mask(_fp(got), 20))
    exp = a.copy(); exp[:5] = 0; exp[15:] = 0
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(got, oo.dynamic_mask(a))
    short = np.ones(3, np.float32)
    N.check(lib.pf_host_online_dynamic_mask(_fp(short), 3))
    assert (short == 0).all()


def test_online_cif_bit_exact_with_carry(lib):
    rng = np.random.default_rng(1)
    D = 512
    carry_a, carry_h = np.float32(0.0), np.zeros(D, np.float32)
    for step in range(6):                                            # six consecutive chunks, state carried
        enc = rng.standard_normal((20, D)).astype(np.float32)
        al = oo.dynamic_mask(rng.uniform(0, 0.6, 20).astype(np.float32))
        h = np.concatenate([carry_h[None], enc]).astype(np.float32)
        a = np.concatenate([[carry_a], al]).astype(np.float32)
        fired = np.zeros((32, D), np.float32)
        nf, ca = C.c_int32(), C.c_float()
        ch = np.zeros(D, np.float32)
        N.check(lib.pf_host_online_cif(_fp(h), _fp(a), 21, D, 1.0, _fp(fired), 32, nf, ca, _fp(ch)))
        rf, ra, rh = oo.cif(h, a, 1.0)
        assert nf.value == rf.shape[0]
        np.testing.assert_array_equal(fired[: nf.value], rf)
        assert np.float32(ca.value) == ra
        np.testing.assert_array_equal(ch, rh)
        carry_a, carry_h = ra, rh
    # known answer: alphas [0.6, 0.6] over hiddens [1.., 2..] -> one fire = 0.6*1 + 0.4*2 = 1.4, carry 0.2 / hidden 2
    h = np.stack([np.full(4, 1, np.float32), np.full(4, 2, np.float32)])
    a = np.asarray([0.6, 0.6], np.float32)
    fired = np.zeros((2, 4), np.float32); nf, ca = C.c_int32(), C.c_float(); ch = np.zeros(4, np.float32)
    N.check(lib.pf_host_online_cif(_fp(h), _fp(a), 2, 4, 1.0, _fp(fired), 2, nf, ca, _fp(ch)))
    assert nf.value == 1 and np.allclose(fired[0], 1.4, atol=1e-6) and abs(ca.value - 0.2) < 1e-6 and np.allclose(ch, 2.0, atol=1e-6)


def test_online_decode_text(lib):
    toks = ["<blank>", "<s>", "</s>", "<unk>", "欢", "迎", "he@@", "llo", "World", "x"]
    arr = (C.c_char_p * len(toks))(*[t.encode() for t in toks])
    for ids in ([0, 0, 4, 5, 6, 7, 8, 2, 9], [0, 0], [3, 8, 8], [0, 0, 6, 7, 4]):
        a = np.asarray(ids, np.int64)
        buf = C.create_string_buffer(256)
        N.check(lib.pf_host_online_decode(arr, len(toks), a.ctypes.data_as(C.POINTER(C.c_int64)), len(ids), buf, 256))
        assert buf.value.decode() == oo.decode_text(toks, ids)
    a = np.asarray([0, 0, 4, 5, 6, 7, 8, 2, 9], np.int64)
    buf = C.create_string_buffer(256)
    N.check(lib.pf_host_online_decode(arr, len(toks), a.ctypes.data_as(C.POINTER(C.c_int64)), 9, buf, 256))
    assert buf.value.decode() == "欢迎hello world"                   # lower-cased, </s> stops, @@ joins


def test_oracle_stream_chunking():
    """The constructor queues a chunk of silence; every AddSamples releases at most ONE 9600-sample chunk; the first
    chunk carries a repeated first frame (61 fbank frames) so 60 are consumed and one is left over."""
    from aliparaformerasr_amd import weights as W
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=32)
    rec = oo.OnlineRecognizer(cfg, W.synth_weights(cfg, 1), W.synth_cmvn(), ["t%d" % i for i in range(32)], quant="fp32")
    s = rec.create_stream()
    assert s.get_decode_chunk() is None
    s.add_samples(np.zeros(1, np.float32))                           # 9601 cached -> the silence chunk goes through
    assert s.speech.shape == (61, 80)
    c = s.get_decode_chunk()
    assert c.shape == (20, 560) and (c[:10] == 0).all() and s.speech.shape == (1, 80) and s.start_idx == 10
    s.add_samples(W.synth_audio(30000, 1))                           # 30001 cached -> one chunk, 20401 stay
    assert s.speech.shape == (61, 80) and len(s.cache_samples) == 20401
    c2 = s.get_decode_chunk()
    np.testing.assert_array_equal(c2[:10], c[10:])                   # the 10-frame feature cache
    assert s.start_idx == 20

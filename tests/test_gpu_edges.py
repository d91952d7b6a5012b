"""GPU: edge cases of the offline path — shapes the headline benchmark never visits."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import model as om

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def _cmp(res, ref, tol=TOL):
    assert res.logits.shape == ref["logits"].shape
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    err = np.abs(res.logits - ref["logits"]).max() if ref["logits"].size else 0.0
    assert err < tol, err


@pytest.fixture(scope="module")
def small():
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=200)
    w = W.synth_weights(cfg, seed=31)
    w["predictor.out.bias"] = np.asarray([-0.4], np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    orc = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16")
    yield eng, orc, cfg, w, cmvn
    eng.close()


def test_very_ragged_batch_with_all_pad_row(small):
    """0.5 s next to 6 s: most rows of the short utterance are PadSequence sentinel rows (LayerNorm on
    |x| ~ 1.7e7); an utterance shorter than one LFR frame contributes only sentinel rows."""
    eng, orc, cfg, w, cmvn = small
    audio = [W.synth_audio(n, 500 + u) for u, n in enumerate((96000, 8000, 700, 51000))]
    speech = _speech(audio, cmvn)
    assert speech.shape[1] == 100 and fe.wav_frontend(audio[2], fe.FrontendConf(dither=0.0), *cmvn).shape[0] == 0
    ref = orc.paraformer(speech)
    res = eng.recognize(audio, want_logits=True)
    _cmp(res, ref)


def test_single_frame_and_tiny_inputs(small):
    eng, orc, cfg, w, cmvn = small
    for n in (960, 1100, 3000):                       # T = 1, 1, 3 LFR frames
        a = [W.synth_audio(n, 600 + n)]
        speech = _speech(a, cmvn)
        ref = orc.paraformer(speech)
        res = eng.recognize(a, want_logits=True)
        assert res.L == ref["logits"].shape[1]
        _cmp(res, ref)


def test_audio_shorter_than_one_lfr_frame_is_an_error(small):
    from aliparaformerasr_amd._native import PfError
    eng = small[0]
    with pytest.raises(PfError):
        eng.recognize([W.synth_audio(500, 1)])


def test_long_utterance_many_key_tiles(small):
    """60 s: T = 1000 -> 8 query tiles x 16 key tiles per head; decoder L ~ 180."""
    eng, orc, cfg, w, cmvn = small
    a = [W.synth_audio(960000, 77)]
    speech = _speech(a, cmvn)
    assert speech.shape[1] == 1000
    ref = orc.paraformer(speech)
    res = eng.recognize(a, want_logits=True)
    _cmp(res, ref, 3e-2)


def test_no_fire_gives_empty_hypotheses():
    """alphas ~ 0 everywhere: sum = tail threshold 0.45 < 1 -> token_num = 0, L = 0, no decoder launch."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64)
    w = W.synth_weights(cfg, seed=8)
    w["predictor.out.bias"] = np.asarray([-30.0], np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    res = eng.recognize([W.synth_audio(16000, 1), W.synth_audio(12000, 2)])
    assert res.L == 0 and res.token_ids.shape == (2, 0) and list(res.token_num) == [0, 0]
    eng.close()


def test_timestamp_head_batch_larger_than_one_lstm_tile():
    """B = 40 > 32: the BiLSTM step kernel runs two utterance tiles; peaks against the oracle."""
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=64, timestamp_head=True)
    w = W.synth_weights(cfg, seed=12)
    w["predictor.out.bias"] = np.asarray([0.0], np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    audio = [W.synth_audio(9600, 800 + u) for u in range(40)]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").paraformer(speech)
    res = eng.recognize(audio)
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    d = np.abs(res.cif_peak - ref["us_cif_peak"])
    d = np.minimum(d, np.abs(d - 0.9999))
    assert np.quantile(d, 0.99) < 2e-2
    for b in (0, 31, 32, 39):
        f_dev = np.nonzero(res.cif_peak[b] > 1 - 1e-4)[0]
        f_ref = np.nonzero(ref["us_cif_peak"][b] > 1 - 1e-4)[0]
        assert len(f_dev) == len(f_ref) and np.all(np.abs(f_dev - f_ref) <= 1)
    eng.close()


def test_concurrent_callers_on_one_engine_are_serialised(small):
    """The reference serialises AddSamples with a static lock and allows concurrent GetResults
    (OfflineStream.cs:19; SURVEY 8b 'Threading'): calls on one handle from several threads must be safe and
    give the single-threaded answers."""
    import threading
    eng = small[0]
    audios = [[W.synth_audio(20000 + 3000 * k, 900 + k), W.synth_audio(9000 + 500 * k, 950 + k)] for k in range(4)]
    expect = [eng.recognize(a, want_logits=True) for a in audios]
    got = [None] * 4
    errs = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = eng.recognize(audios[k], want_logits=True)
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in range(4):
        assert np.array_equal(got[k].token_ids, expect[k].token_ids)
        assert np.array_equal(got[k].logits, expect[k].logits)


def test_engine_lifecycle_releases_device_memory():
    """Create / use / destroy engines repeatedly (each with growing and shrinking shapes): device memory
    returns to its starting level (grow-only arenas, hipGraph, per-thread slots are all owned by the engine)."""
    import torch
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=1, vocab=64, timestamp_head=True)
    w = W.synth_weights(cfg, 3)
    w["predictor.out.bias"] = np.asarray([0.0], np.float32)
    blob = W.pack_pfw(cfg, w)
    cmvn = W.synth_cmvn()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(4):
        eng = Engine(weights=blob, cmvn=cmvn, device=0)
        for n, B in ((16000, 1), (160000, 6), (8000, 2), (64000, 3)):
            r = eng.recognize([W.synth_audio(n, 10 * rep + u) for u in range(B)])
            assert r.token_ids.shape[0] == B
        eng.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free0 - free1) < 64 << 20, (free0, free1)


def test_malformed_weight_containers_fail_loudly():
    """Truncated / corrupted / incomplete PFW containers must raise, never crash or half-load."""
    from aliparaformerasr_amd._native import PfError
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=32)
    w = W.synth_weights(cfg, 4)
    good = W.pack_pfw(cfg, w)
    cmvn = W.synth_cmvn()
    bad = [b"", b"PFW1", good[:100], good[: len(good) // 2], b"XXXX" + good[4:], good[:16] + b"{" * 40 + good[56:]]
    w2 = dict(w); w2.pop("encoder.layers.0.ffn.w1.bias")
    bad.append(W.pack_pfw(cfg, w2))                                   # missing tensor
    w3 = dict(w); w3["decoder.output.weight"] = w3["decoder.output.weight"][:-1]
    bad.append(W.pack_pfw(cfg, w3))                                   # vocab mismatch
    bad.append(W.pack_pfw(dict(cfg, d_model=256), w))                 # unsupported geometry
    for blob in bad:
        with pytest.raises(PfError):
            Engine(weights=np.frombuffer(blob, np.uint8) if blob else np.zeros(0, np.uint8), cmvn=cmvn, device=0)
    Engine(weights=good, cmvn=cmvn, device=0).close()                 # the device is still usable afterwards
    with pytest.raises(PfError):
        Engine(weights=good, cmvn=(cmvn[0][:100], cmvn[1][:100]), device=0).recognize([W.synth_audio(16000, 1)])


def test_five_minute_utterance(small):
    """T = 5000 LFR frames: 40 query tiles x 79 key tiles per head, L ~ 900 tokens, CIF scan over 5001 frames."""
    eng, orc, cfg, w, cmvn = small
    a = [W.synth_audio(4800000, 123)]
    speech = _speech(a, cmvn)
    assert speech.shape[1] == 5000
    ref = orc.paraformer(speech)
    res = eng.recognize(a, want_logits=True)
    assert res.L == ref["logits"].shape[1] > 500
    _cmp(res, ref, 3e-2)


def test_handles_are_idempotent_and_answer_disposed():
    """pf_engine_destroy twice (Dispose() + finaliser, OfflineRecognizer.cs:448-476) and calls on a destroyed
    handle: PF_ERR_DISPOSED, never a crash."""
    import ctypes as C
    from aliparaformerasr_amd import _native as N
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=32)
    eng = Engine(weights=W.pack_pfw(cfg, W.synth_weights(cfg, 4)), cmvn=W.synth_cmvn(), device=0)
    h = eng._h
    lib = N.load()
    lib.pf_engine_destroy(h)
    lib.pf_engine_destroy(h)                     # second destroy of the same handle
    t = C.c_int32()
    assert lib.pf_frontend_num_frames(h, 16000, t) == N.PF_ERR_DISPOSED
    assert lib.pf_sync(h) == N.PF_ERR_DISPOSED
    eng._h = None


def test_malformed_containers_round2():
    """Containers that used to reach the device or the stack: mismatched LayerNorm width, predictor.out sizes,
    negative dims, a header nested 10 000 deep, a header length that wraps, unsupported framing."""
    import json
    import struct
    from aliparaformerasr_amd._native import PfError, PF_ERR_FORMAT, PF_ERR_UNSUPPORTED
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=1, dec_layers=1, vocab=32)
    w = W.synth_weights(cfg, 4)
    good = W.pack_pfw(cfg, w)
    cmvn = W.synth_cmvn()

    def fails(blob, code=PF_ERR_FORMAT, **kw):
        with pytest.raises(PfError) as ei:
            Engine(weights=np.frombuffer(blob, np.uint8), cmvn=cmvn, device=0, **kw)
        assert ei.value.code == code, ei.value

    for name, cut in (("encoder.layers.0.norm2.weight", 100), ("decoder.layers.0.ffn.norm.weight", 512),
                      ("predictor.out.weight", 100), ("decoder.after_norm.bias", 511)):
        w2 = dict(w)
        w2[name] = w2[name].reshape(-1)[:cut].copy()
        if name.endswith(".weight") and "norm" in name:
            w2[name.replace(".weight", ".bias")] = w2[name.replace(".weight", ".bias")][:cut].copy()
        if name.endswith(".bias") and "norm" in name:
            w2[name.replace(".bias", ".weight")] = w2[name.replace(".bias", ".weight")][:cut].copy()
        fails(W.pack_pfw(cfg, w2))
    # header surgery
    hlen = struct.unpack("<Q", good[8:16])[0]
    hdr = json.loads(good[16:16 + hlen])
    def rebuild(h):
        hb = json.dumps(h, separators=(",", ":")).encode().replace(b"Infinity", b"1e400")
        old_off = (16 + hlen + 255) // 256 * 256
        new_off = (16 + len(hb) + 255) // 256 * 256
        return good[:8] + struct.pack("<Q", len(hb)) + hb + b"\0" * (new_off - 16 - len(hb)) + good[old_off:]
    Engine(weights=np.frombuffer(rebuild(hdr), np.uint8), cmvn=cmvn, device=0).close()   # the surgery itself is sound
    h2 = json.loads(json.dumps(hdr)); h2["tensors"][0]["shape"][0] = -h2["tensors"][0]["shape"][0]
    fails(rebuild(h2))
    fails(good[:8] + struct.pack("<Q", (1 << 64) - 8) + good[16:])           # 16 + hlen wraps to 8
    deep = b"[" * 10000
    fails(good[:8] + struct.pack("<Q", len(deep)) + deep + good[16:])
    # integer fields must be integers in range (round 3: 1e400 used to be cast from inf)
    for key, val in (("d_model", 1e400), ("vocab", 0.5), ("enc_layers", 3e9), ("kernel", -1e300)):
        h3 = json.loads(json.dumps(hdr)); h3["config"][key] = val
        fails(rebuild(h3))
    h3 = json.loads(json.dumps(hdr)); h3["tensors"][1]["offset"] = 256.5
    fails(rebuild(h3))
    h3 = json.loads(json.dumps(hdr)); h3["tensors"][1]["dtype"] = "i4"
    fails(rebuild(h3))
    fails(good, code=PF_ERR_UNSUPPORTED, frame_length_ms=20)
    fails(good, code=PF_ERR_UNSUPPORTED, frame_shift_ms=5)
    Engine(weights=good, cmvn=cmvn, device=0, frame_length_ms=25, frame_shift_ms=10).close()


def test_result_slot_is_released_by_the_fetch_that_delivers_ids(small):
    """learn-L-then-fetch: the forward publishes a per-thread slot (it may carry a B*L*V host copy of the
    log-probs), the pf_fetch that receives token_ids releases it; a thread that dies takes its slot with it."""
    import threading
    eng = small[0]
    a = [W.synth_audio(32000, 5)]
    r1 = eng.recognize(a, want_logits=True)
    r2 = eng.recognize(a, want_logits=True)
    np.testing.assert_array_equal(r1.logits, r2.logits)
    out = {}
    def work():
        out["r"] = eng.recognize(a, want_logits=True)
    ts = [threading.Thread(target=work) for _ in range(3)]
    for t in ts:
        t.start(); t.join()                                # three short-lived threads, possibly one re-used OS id
    np.testing.assert_array_equal(out["r"].logits, r1.logits)

"""GPU: the reference's own API-contract tests (AliParaformerAsr.Tests/OfflineRecognizerTests .cs:166-353)
replayed through the drop-in OfflineRecognizer / OfflineStream mirror, plus an end-to-end
text check against the oracle (front-end -> model -> arg-max -> DecodeMulti)."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu

VOCAB = 300


def _tokens():
    toks = ["<blank>", "<s>", "</s>", "<unk>"]
    cjk = [chr(0x4E00 + 37 * i) for i in range(120)]
    bpe = []
    for i in range(VOCAB - 4 - len(cjk)):
        w = "w%d" % i
        bpe.append(w + "@@" if i % 3 == 0 else ("▁" + w if i % 3 == 1 else w))
    return toks + cjk + bpe


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("model")
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=VOCAB)
    w = W.synth_weights(cfg, seed=77)
    W.save_pfw(str(d / "model.pfw"), cfg, w)
    shift, scale = W.synth_cmvn()
    (d / "am.mvn").write_text(fe.format_mvn_text(shift, scale))
    (d / "asr.yaml").write_text("model: paraformer\nuse_itn: false\nfrontend_conf:\n  fs: 16000\n  window: hamming\n"
                                "  n_mels: 80\n  dither: 0\n  lfr_m: 7\n  lfr_n: 6\n  snip_edges: false\n")
    (d / "tokens.txt").write_text("\n".join(_tokens()) + "\n", encoding="utf-8")
    (d / "hotword.txt").write_text(_tokens()[10] + _tokens()[20] + "\n", encoding="utf-8")
    return d, cfg, w, (shift, scale)


def _make(model_dir, **kw):
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    d = model_dir[0]
    return OfflineRecognizer(modelFilePath=str(d / "model.pfw"), configFilePath=str(d / "asr.yaml"),
                             mvnFilePath=str(d / "am.mvn"), tokensFilePath=str(d / "tokens.txt"), **kw)


def test_init_with_valid_params(model_dir):                       # :167
    assert _make(model_dir) is not None


def test_init_with_missing_tokens_file(model_dir):                # :185
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer, RecognizerException
    d = model_dir[0]
    with pytest.raises(RecognizerException, match="tokens invalid"):
        OfflineRecognizer(str(d / "model.pfw"), str(d / "asr.yaml"), str(d / "am.mvn"), "")


def test_create_stream_add_samples(model_dir):                    # :214 — 1 s of zeros accepted
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    s.AddSamples(np.zeros(16000, np.float32))
    assert s is not None and s.SpeechLength == 16 * 560


def test_get_result_with_valid_stream(model_dir):                 # :232 — silence gives a well-formed entity
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    s.AddSamples(np.zeros(16000, np.float32))
    res = r.GetResult(s)
    assert res.Text is not None and res.Tokens is not None and res.Timestamps is not None
    assert res.TextLen == len(res.Text.encode("utf-16-le")) // 2


def test_add_samples_with_valid_samples(model_dir):               # :267 — 1000 x 0.1f must not throw
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    s.AddSamples(np.full(1000, 0.1, np.float32))
    assert s.SpeechLength == 560                                  # 6 fbank frames -> 1 LFR frame


def test_add_samples_with_null_samples(model_dir):                # :286 — ArgumentNullException("source")
    from aliparaformerasr_amd.offline_recognizer import ArgumentNullException
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    with pytest.raises(ArgumentNullException) as ei:
        s.AddSamples(None)
    assert ei.value.ParamName == "source"


def test_set_hotwords(model_dir):                                 # :303 / :322
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    assert s.Hotwords == []
    s.Hotwords = [[5, 6, 7], [9], []]
    assert s.Hotwords == [[5, 6, 7], [9], []]
    s.Hotwords = None
    assert s.Hotwords is None


def test_dispose_releases_resources(model_dir):                   # :341
    from aliparaformerasr_amd.offline_recognizer import ObjectDisposedException
    r = _make(model_dir)
    r.Dispose()
    with pytest.raises(ObjectDisposedException) as ei:
        r.CreateOfflineStream()
    assert ei.value.ObjectName == "OfflineRecognizer"
    r.Dispose()                                                   # idempotent (Dispose(bool) pattern)


def test_empty_stream_list_is_noop(model_dir):                    # OfflineRecognizer.cs:120-123
    assert _make(model_dir).GetResults([]) == []


def test_stream_without_samples_fails_like_reference(model_dir):  # PadSequence NRE inside Forward's try
    from aliparaformerasr_amd.offline_recognizer import RecognizerException
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    with pytest.raises(RecognizerException, match="Offline recognition failed"):
        r.GetResult(s)


def test_end_to_end_text_matches_oracle(model_dir):
    d, cfg, w, cmvn = model_dir
    r = _make(model_dir, hotwordFilePath=str(d / "hotword.txt"))
    audio = [W.synth_audio(n, 40 + u) for u, n in enumerate((40000, 48000, 33000))]
    streams = []
    for a in audio:
        s = r.CreateOfflineStream()
        s.AddSamples(a)
        streams.append(s)
    results = r.GetResults(streams)
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(len(audio), T, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").paraformer(speech)
    ids_ref = om.argmax_last(ref["logits"])
    srt = np.sort(ref["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 0.04
    table = _tokens()
    for b, (s, res) in enumerate(zip(streams, results)):
        ids = np.asarray(s.Tokens)
        assert ids.shape[0] == ids_ref.shape[1]
        assert (ids[safe[b]] == ids_ref[b][safe[b]]).all()
        # DecodeMulti over the ids the device produced must equal the oracle's DecodeMulti
        L = ids.shape[0]
        text, tlen, toks, ts = glue.decode_multi_one(table, ids.tolist(), [[0, 0]] * L)
        assert (res.Text, res.TextLen, res.Tokens, res.Timestamps) == (text, tlen, toks, ts)
        assert s.SpeechLength == 0                                  # RemoveChunk after > 2 tokens
    # a second GetResults on a consumed stream fails like the reference (Speech == null)
    from aliparaformerasr_amd.offline_recognizer import RecognizerException
    with pytest.raises(RecognizerException):
        r.GetResults(streams[:1])


def test_stream_outlives_its_recognizer(model_dir):
    """A stream handle used after its recognizer was freed answers ObjectDisposedException — it does not
    dereference freed memory (the stream shares ownership of the recognizer object, not of the device engine)."""
    import gc
    from aliparaformerasr_amd.offline_recognizer import ObjectDisposedException
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    s.AddSamples(np.zeros(16000, np.float32))
    s._recognizer = None                           # drop the Python-side keep-alive: the native side must cope alone
    r._lib.pf_recognizer_free(r._h)
    r._h = None
    del r
    gc.collect()
    with pytest.raises(ObjectDisposedException) as ei:
        s.AddSamples(np.zeros(16000, np.float32))
    assert ei.value.ObjectName == "OfflineRecognizer"
    assert s.SpeechLength > 0                      # host-side state is still readable
    s.Dispose()
    with pytest.raises(ObjectDisposedException) as ei:
        s.SpeechLength
    assert ei.value.ObjectName == "OfflineStream"
    s.Dispose()                                    # idempotent


def test_disposed_stream_answers_object_disposed(model_dir):
    from aliparaformerasr_amd.offline_recognizer import ObjectDisposedException
    r = _make(model_dir)
    s = r.CreateOfflineStream()
    r.DisposeOfflineStream(s)
    for call in (lambda: s.AddSamples(np.zeros(100, np.float32)), lambda: s.Tokens, lambda: s.Hotwords):
        with pytest.raises(ObjectDisposedException) as ei:
            call()
        assert ei.value.ObjectName == "OfflineStream"
    # a recognizer that creates one stream per utterance does not accumulate them
    for _ in range(200):
        t = r.CreateOfflineStream()
        t.AddSamples(np.zeros(1600, np.float32))
        r.DisposeOfflineStream(t)
    r.Dispose()


def test_conf_frame_length_and_shift_are_ignored_like_the_reference(model_dir, tmp_path):
    """WavFrontend.cs:21-27 builds OnlineFbank from dither / snip_edges / window / fs / n_mels only: `frame_length` and
    `frame_shift` of the yaml never reach it, so a conf with other values runs in the reference at the kaldi defaults
    (25 ms / 10 ms).  The mirror must do the same — not refuse the file — and produce the same features."""
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    d = model_dir[0]
    odd = tmp_path / "asr_odd.yaml"
    odd.write_text((d / "asr.yaml").read_text() + "  frame_length: 20\n  frame_shift: 5\n")
    a = _make(model_dir)
    b = OfflineRecognizer(modelFilePath=str(d / "model.pfw"), configFilePath=str(odd), mvnFilePath=str(d / "am.mvn"),
                          tokensFilePath=str(d / "tokens.txt"))
    x = W.synth_audio(32000, 5)
    sa, sb = a.CreateOfflineStream(), b.CreateOfflineStream()
    sa.AddSamples(x)
    sb.AddSamples(x)
    assert sa.SpeechLength == sb.SpeechLength == 33 * 560
    assert a.GetResult(sa).Text == b.GetResult(sb).Text


@pytest.mark.timeout(300)
def test_stream_handles_are_quarantined_then_recycled(model_dir):
    """One stream per utterance for the life of a server: a freed handle answers PF_ERR_DISPOSED (double free / use after
    free, the Dispose-then-finaliser pattern of OfflineRecognizer.cs:448-476) for as long as it sits in the quarantine, and
    its shell is handed out again after 65 536 younger frees — the process does not grow by one shell per stream."""
    import ctypes as C
    from aliparaformerasr_amd import _native as N
    r = _make(model_dir)
    lib = N.load()
    first = C.c_void_p()
    N.check(lib.pf_recognizer_create_stream(r._h, C.byref(first)))
    lib.pf_stream_free(first)
    lib.pf_stream_free(first)                                   # second free of the same handle: no-op
    n = C.c_int32()
    assert lib.pf_stream_num_feature_floats(first, C.byref(n)) == N.PF_ERR_DISPOSED
    seen = set()
    recycled = False
    for i in range(66000):
        h = C.c_void_p()
        N.check(lib.pf_recognizer_create_stream(r._h, C.byref(h)))
        if h.value == first.value:
            recycled = True
            assert lib.pf_stream_num_feature_floats(h, C.byref(n)) == 0 and n.value == 0     # a fresh, live stream
        assert h.value not in seen or i > 65536
        seen.add(h.value)
        lib.pf_stream_free(h)
        if i < 65000:
            assert lib.pf_stream_num_feature_floats(first, C.byref(n)) == N.PF_ERR_DISPOSED or recycled
    assert recycled, "the first freed shell was never handed out again"
    assert len(seen) <= 65536 + 2
    r.Dispose()

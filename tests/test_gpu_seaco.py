"""GPU: SeACo-paraformer (BASELINE.json configs[4], SURVEY.md §8a rows 11e + 16): hotword embedder, bias
decoder, NO-BIAS merge, with the BiCIF timestamp head, against the oracle's restatement of the export
(oracle/model.py::seaco — parity unpinned: the graph itself is external to the reference)."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu
VOCAB, NOBIAS = 120, 111


def _model(seed=21):
    cfg = W.seaco_paraformer_config(enc_layers=2, dec_layers=2, vocab=VOCAB, seaco_layers=2, seaco_nobias=NOBIAS)
    w = W.synth_weights(cfg, seed)
    w["predictor.out.bias"] = np.asarray([0.0], np.float32)
    # random heads never pick NO-BIAS: lift its logit so that roughly half of the positions keep the ASR row
    w["seaco.output.bias"][NOBIAS] += 2.6
    return cfg, w


@pytest.fixture(scope="module")
def seaco_setup():
    from aliparaformerasr_amd.engine import Engine
    cfg, w = _model()
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    audio = [W.synth_audio(n, 300 + u) for u, n in enumerate((40000, 31000, 36000))]
    conf = fe.FrontendConf(dither=0.0)
    speech = fe.pad_sequence([fe.wav_frontend(a, conf, *cmvn) for a in audio]).reshape(len(audio), -1, 560)
    hw = np.asarray(glue.pad_list([[5, 6, 7], [9, 10], [30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41], [1]]), np.int32)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").seaco(speech, hw)
    yield eng, cfg, w, cmvn, audio, speech, hw, ref
    eng.close()


def test_seaco_logits_and_ids_vs_oracle(seaco_setup):
    eng, cfg, w, cmvn, audio, speech, hw, ref = seaco_setup
    r = eng.forward_feats(speech, want_logits=True, hotwords=hw)
    assert np.array_equal(r.token_num, ref["token_num"])
    assert r.logits.shape == ref["logits"].shape
    # which rows took the hotword distribution: compare the decision where the oracle's NO-BIAS margin is clear
    dha = ref["dha_logits"]
    top = np.sort(dha, axis=-1)
    nob_ref = np.argmax(dha, -1) == NOBIAS
    other = np.where(np.arange(VOCAB)[None, None, :] == NOBIAS, -np.inf, dha).max(-1)
    clear = np.abs(dha[..., NOBIAS] - other) > 0.05
    assert 0.15 < nob_ref.mean() < 0.85, nob_ref.mean()          # the test exercises both branches
    err = np.abs(r.logits - ref["logits"]).max(-1)
    assert err[clear].max() < 3e-2, err[clear].max()
    ids_ref = om.argmax_last(ref["logits"])
    srt = np.sort(ref["logits"], axis=-1)
    safe = clear & ((srt[..., -1] - srt[..., -2]) > 0.06)
    assert safe.mean() > 0.5
    np.testing.assert_array_equal(r.token_ids[safe], ids_ref[safe])
    # timestamps head still present
    assert r.cif_peak is not None and r.cif_peak.shape == ref["us_cif_peak"].shape


def test_seaco_no_hotwords_equals_asr_branch(seaco_setup):
    """bias_embed [B,0,512] (OfflineProjOfSeacoParaformer.cs:85-86): the bias branch has nothing to attend;
    the build returns the ASR rows unchanged."""
    eng, cfg, w, cmvn, audio, speech, hw, ref = seaco_setup
    r = eng.forward_feats(speech, want_logits=True, hotwords=None)
    err = np.abs(r.logits - ref["asr_logits"]).max()
    assert err < 3e-2, err
    a = eng.recognize(audio, hotwords=hw)
    b = eng.forward_feats(speech, hotwords=hw)
    assert np.array_equal(a.token_ids, b.token_ids)


def test_seaco_recognizer_hotwords(tmp_path, seaco_setup):
    """Through the recognizer mirror: constructor hotword file (+[1] terminator) is the default, a stream's own
    Hotwords override it (OfflineProjOfSeacoParaformer.cs:52-60); a null list fails the recognition."""
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer, RecognizerException
    eng, cfg, w, cmvn, audio, speech, hw, ref = seaco_setup
    d = tmp_path
    W.save_pfw(str(d / "model.pfw"), cfg, w)
    (d / "am.mvn").write_text(fe.format_mvn_text(*cmvn))
    toks = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + 3 * i) for i in range(VOCAB - 4)] + ["<unk>"]
    (d / "tokens.txt").write_text("\n".join(toks) + "\n", encoding="utf-8")
    (d / "asr.yaml").write_text("model: seacoparaformer\nfrontend_conf:\n  dither: 0.0\n")
    (d / "hotword.txt").write_text(toks[5] + toks[6] + toks[7] + "\n" + toks[9] + toks[10] + "\n", encoding="utf-8")
    rec = OfflineRecognizer(str(d / "model.pfw"), str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"),
                            hotwordFilePath=str(d / "hotword.txt"))
    streams = []
    for a in audio:
        s = rec.CreateOfflineStream()
        s.AddSamples(a)
        streams.append(s)
    rec.GetResults(streams)
    default_hw = np.asarray(glue.pad_list([[5, 6, 7], [9, 10], [1]]), np.int32)
    exp = eng.forward_feats(speech, hotwords=default_hw)
    for b, s in enumerate(streams):
        assert list(s.Tokens) == [int(x) for x in exp.token_ids[b]]
    # per-stream hotwords replace the default list
    s = rec.CreateOfflineStream()
    s.AddSamples(audio[0])
    s.Hotwords = [[30, 31], [1]]
    rec.GetResults([s])
    sp0 = fe.pad_sequence([fe.wav_frontend(audio[0], fe.FrontendConf(dither=0.0), *cmvn)]).reshape(1, -1, 560)
    exp1 = eng.forward_feats(sp0, hotwords=np.asarray(glue.pad_list([[30, 31], [1]]), np.int32))
    assert list(s.Tokens) == [int(x) for x in exp1.token_ids[0]]
    s2 = rec.CreateOfflineStream()
    s2.AddSamples(audio[0])
    s2.Hotwords = None
    with pytest.raises(RecognizerException, match="Offline recognition failed"):
        rec.GetResults([s2])
    rec.Dispose()


def test_hotword_side_is_cached_per_list_not_per_engine(seaco_setup):
    """The embedder output and the bias rows' K / V are computed once per hot-word LIST (round 3) — the reference re-runs
    model_eb every call (OfflineProjOfSeacoParaformer.cs:83-111) with the same result.  A different list, a longer list
    (workspace growth) and the first list again must each give what a fresh engine gives."""
    from aliparaformerasr_amd.engine import Engine
    eng, cfg, w, cmvn, audio, speech, hw, ref = seaco_setup
    hw2 = np.asarray(glue.pad_list([[40, 41], [5, 6, 7], [1]]), np.int32)
    hw3 = np.asarray(glue.pad_list([[i, i + 1, i + 2] for i in range(3, 90, 3)] + [[1]]), np.int32)      # 30 hot words
    fresh = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    want = {}
    for name, h in (("hw3", hw3), ("hw2", hw2), ("hw", hw)):
        want[name] = fresh.forward_feats(speech, want_logits=True, hotwords=h)
        fresh.close()
        fresh = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    fresh.close()
    for name, h in (("hw", hw), ("hw", hw), ("hw2", hw2), ("hw3", hw3), ("hw", hw), ("hw2", hw2)):
        r = eng.forward_feats(speech, want_logits=True, hotwords=h)
        np.testing.assert_array_equal(r.token_ids, want[name].token_ids)
        np.testing.assert_array_equal(r.logits, want[name].logits)
    assert not np.array_equal(want["hw"].logits, want["hw2"].logits)


def test_recognizer_on_the_reference_default_int8_file_name(tmp_path, seaco_setup):
    """The reference CLI's default is paraformer-seaco-large-zh-timestamp with `-accuracy int8`
    (Examples/Program.cs:98-101, Examples/OfflineAliParaformerAsrRecognizer.cs:17-21): a container named
    model.int8.pfw selects math_mode 2 and must RUN for a SeACo + timestamp model — tokens equal to the int8 engine's,
    timestamps present."""
    from aliparaformerasr_amd.engine import Engine
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    _, cfg, w, cmvn, audio, speech, hw, ref = seaco_setup
    d = tmp_path
    W.save_pfw(str(d / "model.int8.pfw"), cfg, w)
    (d / "am.mvn").write_text(fe.format_mvn_text(*cmvn))
    toks = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + 3 * i) for i in range(VOCAB - 4)] + ["<unk>"]
    (d / "tokens.txt").write_text("\n".join(toks) + "\n", encoding="utf-8")
    (d / "asr.yaml").write_text("model: seacoparaformer\nfrontend_conf:\n  dither: 0.0\n")
    (d / "hotword.txt").write_text(toks[5] + toks[6] + toks[7] + "\n" + toks[9] + toks[10] + "\n", encoding="utf-8")
    rec = OfflineRecognizer(str(d / "model.int8.pfw"), str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"),
                            hotwordFilePath=str(d / "hotword.txt"))
    streams = []
    for a in audio:
        s = rec.CreateOfflineStream()
        s.AddSamples(a)
        streams.append(s)
    results = rec.GetResults(streams)
    eng8 = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=2)
    default_hw = np.asarray(glue.pad_list([[5, 6, 7], [9, 10], [1]]), np.int32)
    # (same device front-end as the recognizer: in int8 mode a 1e-5 difference between the numpy and the device fbank moves
    # uint8 codes, and with them near-tie tokens)
    exp = eng8.recognize(audio, hotwords=default_hw)
    for b, s in enumerate(streams):
        assert list(s.Tokens) == [int(x) for x in exp.token_ids[b]]
        if results[b].Text:
            assert len(results[b].Timestamps) > 0
    eng8.close()
    rec.Dispose()

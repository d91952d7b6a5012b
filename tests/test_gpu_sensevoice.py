"""GPU: SenseVoice-small path (BASELINE.json configs[2]) — SAN-M encoder (+ "tp" blocks) + CTC
head + per-frame arg-max, prompt rows from the reference's own embed table
(AliParaformerAsr/data/embed.onnx -> tests/golden/sensevoice_embed.npy), effective prompt ids
per quirk Q7 (OfflineProjOfSenseVoiceSmall.cs:57-74), no CTC collapse (quirk Q6)."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu
VOCAB = 403          # deliberately not a multiple of 4 (sensevoice-small: 25055)


def _cfg():
    return W.sensevoice_small_config(enc_layers=3, tp_layers=2, vocab=VOCAB)


def test_sensevoice_forward_feats_vs_oracle(sv_embed):
    from aliparaformerasr_amd.engine import Engine
    cfg = _cfg()
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    conf = fe.FrontendConf(dither=0.0)
    audio = [W.synth_audio(n, 70 + u) for u, n in enumerate((32000, 24000))]
    feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, cmvn[0], cmvn[1]), sv_embed, use_itn=True) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(2, T, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").sensevoice(speech)
    res = eng.forward_feats(speech, want_logits=True)
    assert res.logits.shape == ref["logits"].shape == (2, T, VOCAB)
    assert res.L == T                                   # one id per frame, prompt frames included
    err = np.abs(res.logits - ref["logits"]).max()
    assert err < 2e-2, err
    ids_ref = om.argmax_last(ref["logits"])
    srt = np.sort(ref["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 0.04
    np.testing.assert_array_equal(res.token_ids[safe], ids_ref[safe])
    eng.close()


def test_sensevoice_audio_in_prepends_prompt_on_device(sv_embed):
    """pf_recognize (audio in, device front-end) must produce exactly what the feature-in entry point
    produces for host-prepended query rows: same T + 4 length, bit-identical log-probs and ids, for both
    use_itn settings (effective language id 14 / 15, quirk Q7)."""
    from aliparaformerasr_amd.engine import Engine
    cfg = _cfg()
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    cmvn = W.synth_cmvn()
    audio = [W.synth_audio(n, 90 + u) for u, n in enumerate((32000, 21000, 27000))]
    for itn in (True, False):
        eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, use_itn=itn)
        feats = [glue.sensevoice_prepend(eng.frontend(a), sv_embed, use_itn=itn) for a in audio]
        T = max(f.shape[0] for f in feats)
        speech = fe.pad_sequence(feats).reshape(len(audio), T, 560)
        a = eng.recognize(audio, want_logits=True)
        b = eng.forward_feats(speech, want_logits=True)
        assert a.L == b.L == T
        assert np.array_equal(a.token_ids, b.token_ids)
        assert np.array_equal(a.logits, b.logits)
        eng.close()


def test_sensevoice_recognizer_prompt_rows(tmp_path, sv_embed):
    """Host mirror: use_itn -> prompt rows [14,1,2,15] prepended to Speech IN PLACE (quirk Q8)."""
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    cfg = _cfg()
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    cmvn = W.synth_cmvn()
    W.save_pfw(str(tmp_path / "model.pfw"), cfg, w)
    (tmp_path / "am.mvn").write_text(fe.format_mvn_text(*cmvn))
    (tmp_path / "asr.yaml").write_text("model: SenseVoiceSmall\nuse_itn: true\nfrontend_conf:\n  dither: 0\n")
    toks = ["<blank>", "<s>", "</s>", "<unk>"] + ["<|tag%d|>" % i for i in range(20)] + [chr(0x4E00 + i) for i in range(VOCAB - 24)]
    (tmp_path / "tokens.txt").write_text("\n".join(toks) + "\n", encoding="utf-8")
    r = OfflineRecognizer(str(tmp_path / "model.pfw"), str(tmp_path / "asr.yaml"), str(tmp_path / "am.mvn"),
                          str(tmp_path / "tokens.txt"))
    a = W.synth_audio(32000, 5)
    s = r.CreateOfflineStream()
    s.AddSamples(a)
    t_plain = s.SpeechLength // 560
    res = r.GetResult(s)
    conf = fe.FrontendConf(dither=0.0)
    feat = glue.sensevoice_prepend(fe.wav_frontend(a, conf, cmvn[0], cmvn[1]), sv_embed, use_itn=True)
    assert feat.shape[0] == t_plain + 4
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").sensevoice(fe.pad_sequence([feat]).reshape(1, -1, 560))
    ids_ref = om.argmax_last(ref["logits"])[0]
    ids = np.asarray(s.Tokens)
    assert ids.shape == ids_ref.shape                     # T + 4 per-frame ids, no collapse
    srt = np.sort(ref["logits"][0], axis=-1)
    safe = (srt[:, -1] - srt[:, -2]) > 0.04
    np.testing.assert_array_equal(ids[safe], ids_ref[safe])
    text, tlen, tk, ts = glue.decode_multi_one(toks, ids.tolist(), [[0, 0]] * len(ids))
    assert (res.Text, res.TextLen, res.Tokens) == (text, tlen, tk)

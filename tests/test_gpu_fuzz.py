"""GPU: randomised shapes through ONE long-lived engine per model kind (the workspace arenas are grow-only and
re-carved per call, so call ORDER matters: stale bytes of a previous, differently shaped call must never leak into
a result), compared with the oracle each time."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu
TOL = 3e-2


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def test_random_call_sequences_paraformer_with_timestamps():
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=160, timestamp_head=True)
    w = W.synth_weights(cfg, 71)
    w["predictor.out.bias"] = np.asarray([-0.2], np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    orc = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16")
    rng = np.random.default_rng(2024)
    for it in range(10):
        B = int(rng.integers(1, 7))
        lens = [int(rng.integers(1000, 130000)) for _ in range(B)]
        if max(lens) < 960:
            lens[0] = 4000
        audio = [W.synth_audio(n, 1000 * it + u) for u, n in enumerate(lens)]
        speech = _speech(audio, cmvn)
        ref = orc.paraformer(speech)
        res = eng.recognize(audio, want_logits=True)
        np.testing.assert_array_equal(res.token_num, ref["token_num"], err_msg="iter %d lens %s" % (it, lens))
        assert res.logits.shape == ref["logits"].shape
        if ref["logits"].size:
            assert np.isfinite(res.logits).all(), (it, lens)
            err = np.abs(res.logits - ref["logits"]).max()
            assert err < TOL, (it, lens, err)
        # peaks: rows with at least one token (token_num = 0 rows renormalise by 0/x = 0 in both)
        d = np.abs(res.cif_peak - ref["us_cif_peak"])
        d = np.minimum(d, np.abs(d - 0.9999))
        assert np.quantile(d, 0.98) < 3e-2, (it, lens)
    eng.close()


def test_random_call_sequences_seaco_and_sensevoice(sv_embed):
    from aliparaformerasr_amd.engine import Engine
    cmvn = W.synth_cmvn()
    rng = np.random.default_rng(7)
    # SeACo without timestamps, varying hotword lists
    cfg = W.seaco_paraformer_config(enc_layers=1, dec_layers=2, vocab=100, seaco_layers=2, seaco_nobias=91, timestamp_head=False)
    w = W.synth_weights(cfg, 5)
    w["predictor.out.bias"] = np.asarray([-0.2], np.float32)
    w["seaco.output.bias"][91] += 2.8
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    orc = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16")
    for it in range(6):
        B = int(rng.integers(1, 5))
        audio = [W.synth_audio(int(rng.integers(3000, 90000)), 50 * it + u) for u in range(B)]
        speech = _speech(audio, cmvn)
        nh = int(rng.integers(1, 9))
        hw = np.asarray(glue.pad_list([list(rng.integers(3, 99, size=int(rng.integers(1, 13)))) for _ in range(nh)] + [[1]]), np.int32)
        ref = orc.seaco(speech, hw)
        res = eng.recognize(audio, want_logits=True, hotwords=hw)
        np.testing.assert_array_equal(res.token_num, ref["token_num"])
        if ref["logits"].size:
            dha = ref["dha_logits"]
            other = np.where(np.arange(100)[None, None, :] == 91, -np.inf, dha).max(-1)
            clear = np.abs(dha[..., 91] - other) > 0.06
            err = np.abs(res.logits - ref["logits"]).max(-1)
            assert np.isfinite(res.logits).all()
            assert (not clear.any()) or err[clear].max() < TOL, (it, err[clear].max())
    eng.close()
    # SenseVoice, audio in
    cfg = W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=131)
    w = W.synth_weights(cfg, 6)
    w["embed.weight"] = sv_embed.astype(np.float32)
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, use_itn=True)
    orc = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16")
    for it in range(6):
        B = int(rng.integers(1, 6))
        audio = [W.synth_audio(int(rng.integers(1200, 100000)), 80 * it + u) for u in range(B)]
        conf = fe.FrontendConf(dither=0.0)
        feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, *cmvn), sv_embed, use_itn=True) for a in audio]
        T = max(f.shape[0] for f in feats)
        speech = fe.pad_sequence(feats).reshape(B, T, 560)
        ref = orc.sensevoice(speech)
        res = eng.recognize(audio, want_logits=True)
        assert res.logits.shape == ref["logits"].shape
        err = np.abs(res.logits - ref["logits"]).max()
        assert np.isfinite(res.logits).all() and err < TOL, (it, err)
    eng.close()

"""BiCIF timestamp head (SURVEY.md §8 rows 11d + 13): device us_cif_peak vs the oracle, and the
timestamps the recognizer derives from it (OfflineRecognizer.cs:172-183, :200-302)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from aliparaformerasr_amd import weights as W


@pytest.fixture(scope="module")
def ts_setup():
    from aliparaformerasr_amd.engine import Engine
    from oracle import frontend as fe, model as om
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=96, timestamp_head=True)
    w = W.synth_weights(cfg, 5)
    w["predictor.out.bias"] = np.asarray([0.0], np.float32)        # ~5 tokens / 10 frames
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    audio = [W.synth_audio(n, u) for u, n in enumerate((40000, 32000, 23000))]
    conf = fe.FrontendConf(dither=0.0)
    speech = fe.pad_sequence([fe.wav_frontend(a, conf, *cmvn) for a in audio]).reshape(len(audio), -1, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").paraformer(speech)
    yield eng, audio, speech, ref
    eng.close()


def test_us_cif_peak_matches_oracle(ts_setup):
    eng, audio, speech, ref = ts_setup
    r = eng.recognize(audio)
    assert r.cif_peak is not None and r.cif_peak.shape == ref["us_cif_peak"].shape
    assert np.array_equal(r.token_num, ref["token_num"])
    pk, rp = r.cif_peak, ref["us_cif_peak"]
    # running integrate in [0, 2): compare modulo the reset (a fire one frame early/late shifts it by thr)
    d = np.abs(pk - rp)
    d = np.minimum(d, np.abs(d - 0.9999))
    assert np.quantile(d, 0.99) < 3e-3 and d.max() < 1e-2, (np.quantile(d, 0.99), d.max())      # measured 6e-4 / 7e-4
    # fire frames: same count (= token_num up to the tail), each within one 20 ms upsampled frame
    thr, margin = 1 - 1e-4, 5e-3
    n_fire = n_safe = 0
    for b in range(pk.shape[0]):
        f_dev = np.nonzero(pk[b] > thr)[0]
        f_ref = np.nonzero(rp[b] > thr)[0]
        assert len(f_dev) == len(f_ref)
        assert np.all(np.abs(f_dev - f_ref) <= 1)
        # a fire whose crossing clears the threshold by more than the peak tolerance on both sides (the integrate one
        # frame earlier is below thr - margin, at the fire above thr + margin) must land on EXACTLY the oracle's frame
        prev = np.where(f_ref > 0, rp[b][np.maximum(f_ref - 1, 0)], 0.0)
        safe = (rp[b][f_ref] > thr + margin) & (prev < thr - margin)
        assert np.array_equal(f_dev[safe], f_ref[safe])
        n_fire += len(f_ref)
        n_safe += int(safe.sum())
    assert n_safe > 0.8 * n_fire, (n_safe, n_fire)          # the exact check covers most fires, not a corner
    print("timestamp head: %d fires, %d exact by margin, peak error 99%% %.2e max %.2e" % (n_fire, n_safe, np.quantile(d, 0.99), d.max()))


def test_forward_feats_peak_deterministic(ts_setup):
    eng, audio, speech, ref = ts_setup
    a = eng.forward_feats(speech).cif_peak
    b = eng.forward_feats(speech).cif_peak
    assert a is not None and np.array_equal(a, b)


def test_recognizer_timestamps(tmp_path, ts_setup):
    """End to end through the recognizer mirror: Timestamps come from time_stamp_lfr6_onnx on the
    device peaks and match the oracle's glue on the same peaks / ids."""
    from aliparaformerasr_amd.offline_recognizer import OfflineRecognizer
    from oracle import glue
    eng, audio, speech, ref = ts_setup
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=96, timestamp_head=True)
    w = W.synth_weights(cfg, 5)
    w["predictor.out.bias"] = np.asarray([0.0], np.float32)
    cmvn = W.synth_cmvn()
    from oracle import frontend as fe
    d = tmp_path
    W.save_pfw(str(d / "model.pfw"), cfg, w)
    (d / "am.mvn").write_text(fe.format_mvn_text(*cmvn))
    toks = ["<blank>", "<s>", "</s>"] + ["t%d" % i for i in range(3, 95)] + ["<unk>"]
    (d / "tokens.txt").write_text("\n".join(toks) + "\n")
    (d / "asr.yaml").write_text("model: paraformer\nfrontend_conf:\n  fs: 16000\n  n_mels: 80\n  lfr_m: 7\n  lfr_n: 6\n  dither: 0.0\n")
    rec = OfflineRecognizer(str(d / "model.pfw"), str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"))
    streams = []
    for a in audio:
        s = rec.CreateOfflineStream()
        s.AddSamples(a)
        streams.append(s)
    res = rec.GetResults(streams)
    r = eng.recognize(audio)
    for b, e in enumerate(res):
        exp_ts = glue.time_stamp_lfr6_onnx(r.cif_peak[b], r.token_ids[b])
        text, _tlen, _toks, ts = glue.decode_multi_one(toks, [int(x) for x in r.token_ids[b]], exp_ts)
        assert e.Text == text
        assert [list(t) for t in e.Timestamps] == [list(t) for t in ts]
        assert len(e.Timestamps) > 0 and all(t[1] >= t[0] for t in e.Timestamps)
    rec.Dispose()

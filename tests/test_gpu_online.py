"""GPU: the streaming path through the C ABI against oracle/online.py.

  * device seams = the two ONNX sessions of OnlineRecognizer.cs (EncoderProj :49-124, DecoderProj :233-334):
    chunk encoder (enc, CIF weights) and cached decoder (log-probs, ids, FSMN caches);
  * the whole OnlineRecognizer / OnlineStream mirror fed in pieces, several streams batched, in lock-step with the
    oracle's restatement of the same managed glue.
Tolerances as for the offline path (f16 GEMM operands): activations 1e-2, log-probs 2e-2, ids where the oracle's
margin is decisive; the host-side state machine (chunking, caches, CIF) is exact and tested on the CPU."""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import model as om
from oracle import online as oo

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
def model(tmp_path_factory):
    d = tmp_path_factory.mktemp("online")
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=3, vocab=VOCAB)
    w = W.synth_weights(cfg, seed=31)
    w["predictor.out.bias"] = np.asarray([0.8], np.float32)          # ~0.6 per frame: a few tokens per 10-frame chunk
    W.save_pfw(str(d / "model.pfw"), cfg, w)
    shift, scale = W.synth_cmvn()
    (d / "am.mvn").write_text(fe.format_mvn_text(shift, scale))
    (d / "asr.yaml").write_text("model: paraformer\nfrontend_conf:\n  fs: 16000\n  window: hamming\n  n_mels: 80\n"
                                "  dither: 0\n  lfr_m: 7\n  lfr_n: 6\n  snip_edges: false\n")
    (d / "tokens.txt").write_text("\n".join(_tokens()) + "\n", encoding="utf-8")
    return d, cfg, w, (shift, scale)


def _engine(model):
    from aliparaformerasr_amd.engine import Engine
    d, cfg, w, cmvn = model
    return Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)


def test_online_encoder_and_decoder_seams(model):
    d, cfg, w, cmvn = model
    eng = _engine(model)
    g = oo.OnlineGraphs(om.Oracle(om.ModelConfig(**cfg), w, quant="fp16"))
    rng = np.random.default_rng(3)
    B, Tc = 3, 20
    speech = (rng.standard_normal((B, Tc, 560)) * 4).astype(np.float32)
    speech[1, :10] = oo.SENTINEL                                      # a first chunk: cached half = sentinel rows
    enc, al = eng.online_encoder(speech)
    enc_r, al_r = g.encoder(speech)
    assert np.abs(enc - enc_r).max() < 1e-2
    assert np.abs(al - al_r).max() < 3e-3
    # decoder with ragged token counts and non-trivial caches
    L = 4
    emb = rng.standard_normal((B, L, 512)).astype(np.float32)
    lens = np.asarray([4, 2, 0], np.int32)
    for b in range(B):
        emb[b, lens[b]:] = 0
    caches = [rng.standard_normal((B, 512, 10)).astype(np.float32) for _ in range(cfg["dec_layers"])]
    logits, ids, cout = eng.online_decoder(enc_r, emb, lens, caches)
    logits_r, cout_r = g.decoder(enc_r, emb, lens, caches)
    valid = (np.arange(L)[None, :] < lens[:, None])
    assert np.abs(logits - logits_r)[valid].max() < 2e-2
    np.testing.assert_array_equal(ids, om.argmax_last(logits))        # exact on the device's own log-probs
    for l in range(cfg["dec_layers"]):
        assert np.abs(cout[l] - cout_r[l]).max() < 1e-2
        # the cache is the last 10 columns of [old cache | masked new positions]: zeros behind a short utterance
        np.testing.assert_array_equal(cout[l][2][:, :6], caches[l][2][:, 4:])   # no token: shifted by L = 4 ...
        assert (cout[l][2][:, 6:] == 0).all()                                     # ... and filled with masked zeros
    _, ids2, _ = eng.online_decoder(enc_r, emb, lens, caches, want_logits=False)
    np.testing.assert_array_equal(ids2, ids)
    eng.close()


def test_online_recognizer_in_lockstep_with_the_oracle(model):
    from aliparaformerasr_amd.online_recognizer import OnlineRecognizer
    d, cfg, w, cmvn = model
    rec = OnlineRecognizer(str(d / "model.pfw"), "", str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"))
    orc = oo.OnlineRecognizer(cfg, w, cmvn, _tokens(), quant="fp16")
    audio = [W.synth_audio(16000 * 3, 60 + u) for u in range(3)]
    streams = [rec.CreateOnlineStream() for _ in audio]
    ostreams = [orc.create_stream() for _ in audio]
    assert streams[0].Tokens == [0, 0]
    step = 4000                                                       # 0.25 s pieces: a chunk completes every 2-3 calls
    n_calls = 0
    for off in range(0, 16000 * 3, step):
        for u in range(3):
            piece = audio[u][off: off + step] if u != 2 or off < 24000 else None     # stream 2 stops early
            if piece is not None:
                streams[u].AddSamples(piece)
                ostreams[u].add_samples(piece)
        res = rec.GetResults(streams)
        ores = orc.get_results(ostreams)
        n_calls += 1
        assert len(res) == 3
        for u in range(3):
            assert len(streams[u].Tokens) == len(ostreams[u].tokens), (off, u)
    # the oracle's trace: several batched forwards happened, with ragged token counts
    fw = [t for t in orc.trace if t["L"] > 0]
    assert len(fw) >= 4 and n_calls == 12
    total = match = 0
    for u in range(3):
        a, b = np.asarray(streams[u].Tokens), np.asarray(ostreams[u].tokens)
        total += len(a); match += int((a == b).sum())
    assert total > 20 and match / total > 0.8, (match, total)         # near-ties of a random-weight model may flip a few ids
    assert isinstance(res[0].Text, str)
    # a second recognizer replays the same audio to the same tokens (deterministic state machine)
    rec2 = OnlineRecognizer(str(d / "model.pfw"), "", str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"))
    s2 = rec2.CreateOnlineStream()
    for off in range(0, 16000 * 3, step):
        s2.AddSamples(audio[0][off: off + step])
        rec2.GetResults([s2])
    # batch composition changes nothing for a stream's own chunks except the padded token positions it inherits
    assert len(s2.Tokens) <= len(streams[0].Tokens)
    rec.Dispose(); rec2.Dispose()


def test_online_contracts(model):
    from aliparaformerasr_amd.offline_recognizer import ArgumentNullException, ObjectDisposedException
    from aliparaformerasr_amd.online_recognizer import OnlineRecognizer
    d = model[0]
    rec = OnlineRecognizer(str(d / "model.pfw"), "", str(d / "asr.yaml"), str(d / "am.mvn"), str(d / "tokens.txt"))
    s = rec.CreateOnlineStream()
    assert rec.GetResults([]) == []
    assert rec.GetResult(s).Text == ""                               # nothing decodable yet: tokens [0, 0] -> ""
    with pytest.raises(ArgumentNullException):
        s.AddSamples(None)
    s.AddSamples(np.zeros(16000, np.float32))                        # silence in: well-formed, no exception
    rec.GetResult(s)
    s.Dispose()
    with pytest.raises(ObjectDisposedException):
        s.AddSamples(np.zeros(10, np.float32))
    rec.Dispose()
    with pytest.raises(ObjectDisposedException):
        rec.CreateOnlineStream()
    rec.Dispose()

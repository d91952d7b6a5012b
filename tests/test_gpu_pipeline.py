"""GPU: the whole offline path (front-end -> SAN-M encoder -> CIF -> decoder -> arg-max),
called through the C ABI, against the CPU oracle on seeded synthetic weights and audio.

Tolerances (f16 MFMA operands, f32 accumulation, f32 LayerNorm/softmax/CIF):
  * vs the oracle with the SAME 16-bit operand rounding points (quant="fp16"): log-probs
    within 2e-2 abs; this isolates kernel correctness from quantisation.
  * vs the pure-fp32 oracle (what onnxruntime computes): log-probs within 5e-2 abs.
  * token ids: identical wherever the oracle's top-1/top-2 margin exceeds twice the logit
    tolerance (random-weight models have many near-ties; a trained model's margins are
    orders of magnitude larger).  token_num / L must be identical.
"""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import model as om

pytestmark = pytest.mark.gpu

TOL_Q = 2e-2
TOL_F = 5e-2


def _make(cfg, seed):
    from aliparaformerasr_amd.engine import Engine
    w = W.synth_weights(cfg, seed=seed)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0)
    return eng, w, cmvn


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def _check(res, ref, tol):
    assert res.logits.shape == ref["logits"].shape
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    assert res.L == ref["logits"].shape[1]
    err = np.abs(res.logits - ref["logits"])
    assert err.max() < tol, err.max()
    tok_ref = om.argmax_last(ref["logits"])
    srt = np.sort(ref["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 2 * tol
    assert safe.mean() > 0.3
    np.testing.assert_array_equal(res.token_ids[safe], tok_ref[safe])
    return float(err.max()), float((res.token_ids == tok_ref).mean())


def test_small_model_all_entry_points():
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=512)
    eng, w, cmvn = _make(cfg, 21)
    audio = [W.synth_audio(n, u) for u, n in enumerate((48000, 32000, 40000))]   # ragged -> pad + sentinel
    speech = _speech(audio, cmvn)
    orc = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16")
    ref = orc.paraformer(speech)
    # (1) fused fast path: raw audio in
    r1 = eng.recognize(audio, want_logits=True)
    _check(r1, ref, TOL_Q)
    # (2) seam IOfflineProj.ModelProj on oracle features (isolates the model from the fbank)
    r2 = eng.forward_feats(speech, want_logits=True)
    _check(r2, ref, TOL_Q)
    # (3) ModelProj with PadSequence on device from ragged feature buffers
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    r3 = eng.model_proj(feats, want_logits=True)
    np.testing.assert_array_equal(r3.token_ids, r2.token_ids)
    np.testing.assert_allclose(r3.logits, r2.logits, atol=1e-6)
    # (4) ids-only path returns the same ids
    r4 = eng.recognize(audio)
    np.testing.assert_array_equal(r4.token_ids, r1.token_ids)
    # encoder alone
    H = eng.op_encoder(speech)
    assert np.abs(H - ref["H"]).max() < 1e-2
    eng.close()


def test_full_depth_paraformer_large_vs_fp16_and_fp32_oracle():
    """paraformer-large geometry (50 + 16 layers, V = 8404), config 1 size (5 s) x 2."""
    cfg = W.paraformer_large_config()
    eng, w, cmvn = _make(cfg, 42)
    audio = [W.synth_audio(80000, u) for u in range(2)]
    speech = _speech(audio, cmvn)
    res = eng.recognize(audio, want_logits=True)
    mc = om.ModelConfig(**cfg)
    ref_q = om.Oracle(mc, w, quant="fp16").paraformer(speech)
    e1, m1 = _check(res, ref_q, TOL_Q)
    ref_f = om.Oracle(mc, w, quant="fp32").paraformer(speech)
    e2, m2 = _check(res, ref_f, TOL_F)
    print("full depth: err vs fp16-oracle %.3e (tok match %.3f), vs fp32-oracle %.3e (tok match %.3f), L=%d" % (e1, m1, e2, m2, res.L))
    assert abs(eng.last_flops() / 2 - 30.9e9) / 30.9e9 < 0.35     # ~30.9 GFLOP per 5 s utterance at L=25
    eng.close()


def test_determinism_and_batch_independence():
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=1, vocab=256)
    eng, w, cmvn = _make(cfg, 33)
    audio = [W.synth_audio(32000, u) for u in range(4)]
    a = eng.recognize(audio, want_logits=True)
    b = eng.recognize(audio, want_logits=True)
    np.testing.assert_array_equal(a.token_ids, b.token_ids)
    np.testing.assert_array_equal(a.logits, b.logits)            # same launch geometry -> bit identical
    # utterances are independent (equal lengths => no padding interaction): sharding the batch
    # across ranks gives the same ids as one big batch
    lo = eng.recognize(audio[:2], want_logits=True)
    hi = eng.recognize(audio[2:], want_logits=True)
    L = a.L
    for part, sl in ((lo, slice(0, 2)), (hi, slice(2, 4))):
        Lp = part.L
        np.testing.assert_allclose(part.logits[:, :min(L, Lp)], a.logits[sl, :min(L, Lp)], atol=2e-3)
    eng.close()


def test_full_depth_sensevoice_small():
    """sensevoice-small geometry (50 + 20 blocks, CTC over 25055 tokens), audio in, query rows prepended on the
    device: error accumulation through 70 layers vs the oracle with the same / without 16-bit rounding points."""
    from oracle import glue
    cfg = W.sensevoice_small_config(use_itn=True)
    eng, w, cmvn = _make(cfg, 42)
    audio = [W.synth_audio(n, 30 + u) for u, n in enumerate((48000, 40000))]
    conf = fe.FrontendConf(dither=0.0)
    feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, *cmvn), w["embed.weight"], use_itn=True) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(len(audio), T, 560)
    res = eng.recognize(audio, want_logits=True)
    mc = om.ModelConfig(**cfg)
    for quant, tol in (("fp16", TOL_Q), ("fp32", TOL_F)):
        ref = om.Oracle(mc, w, quant=quant).sensevoice(speech)
        assert res.logits.shape == ref["logits"].shape == (2, T, 25055)
        err = np.abs(res.logits - ref["logits"])
        assert err.max() < tol, (quant, err.max())
        srt = np.sort(ref["logits"], axis=-1)
        safe = (srt[..., -1] - srt[..., -2]) > 2 * tol
        np.testing.assert_array_equal(res.token_ids[safe], om.argmax_last(ref["logits"])[safe])
    eng.close()


def test_full_depth_seaco_with_timestamps():
    """configs[4] geometry: paraformer-large + BiCIF head + SeACo bias decoder (4 layers, FFN 1024, k = 21) and the
    hotword embedder, one 5 s utterance, 6 hotwords."""
    from oracle import glue
    cfg = W.seaco_paraformer_config()
    eng, w, cmvn = _make(cfg, 42)
    audio = [W.synth_audio(80000, 3)]
    speech = _speech(audio, cmvn)
    hw = np.asarray(glue.pad_list([[11, 12], [100, 200, 300], [4000, 4001, 4002, 4003], [7, 8], [9], [1]]), np.int32)
    res = eng.recognize(audio, want_logits=True, hotwords=hw)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").seaco(speech, hw)
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    dha = ref["dha_logits"]
    nb = cfg["seaco_nobias"]
    other = np.where(np.arange(dha.shape[-1])[None, None, :] == nb, -np.inf, dha).max(-1)
    clear = np.abs(dha[..., nb] - other) > 0.05                # the NO-BIAS decision is not a near-tie
    err = np.abs(res.logits - ref["logits"]).max(-1)
    assert err[clear].max() < 3e-2, err[clear].max()
    d = np.abs(res.cif_peak - ref["us_cif_peak"])
    d = np.minimum(d, np.abs(d - 0.9999))
    assert np.quantile(d, 0.99) < 2e-2
    eng.close()


@pytest.mark.timeout(600)
def test_zero_copy_length_read_back_equals_the_copies():
    """The decoder length and the per-utterance counts reach the host through pinned memory + an event (Engine::ensure_plan_host,
    export_plan_kernel) instead of through three device-to-host copies on a side stream; PF_PLAN_ZERO_COPY=0 is the old form.  The knob
    is read once per process, so each form runs in a process of its own: same L, token_num and ids in the f16, int8 and exact modes on a ragged batch
    (incl. an utterance too short for a token) and on a second call with another batch size (the host buffer grows)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, numpy as np\n"
        "from aliparaformerasr_amd import weights as W\n"
        "from aliparaformerasr_amd.engine import Engine\n"
        "cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=300)\n"
        "blob = W.pack_pfw(cfg, W.synth_weights(cfg, seed=31))\n"
        "out = []\n"
        "for mode in (0, 2, 3):\n"                      # f16, int8, exact: each has its own read-back site
        "    e = Engine(weights=blob, cmvn=W.synth_cmvn(), device=0, math_mode=mode)\n"
        "    for lens in ((48000, 1200, 80000, 16000, 33000), tuple(16000 + 977 * u for u in range(70))):\n"
        "        r = e.recognize([W.synth_audio(n, 400 + u) for u, n in enumerate(lens)])\n"
        "        out.append({'L': int(r.L), 'token_num': np.asarray(r.token_num).tolist(), 'ids': np.asarray(r.token_ids).tolist()})\n"
        "    e.close()\n"
        "print(json.dumps(out))\n")
    res = {}
    for v in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PF_PLAN_ZERO_COPY=v), cwd=root, capture_output=True,
                           text=True, timeout=280)
        assert p.returncode == 0, p.stderr[-3000:]
        res[v] = json.loads([l for l in p.stdout.splitlines() if l.startswith("[")][-1])
    assert res["0"] == res["1"]
    assert len(res["1"]) == 6 and res["1"][0]["L"] > 0 and len(res["1"][1]["token_num"]) == 70


"""GPU: math_mode = 1, the fp32 parity mode (pf_engine_config.math_mode; DESIGN.md "fp32 parity mode").

The reference runs the exported graphs through onnxruntime in fp32 (OfflineModel.cs:41-57).  The default engine path
feeds the matrix cores f16 operands, which is good for ~1e-2 on log-probs; this mode keeps every activation and every
weight in fp32 (v_mfma_f32_32x32x2_f32, fp32 softmax / LayerNorm / CIF) so that the only differences from the fp32
oracle are summation order.  Tolerance: 2e-4 abs on log-probs of magnitude ~10 (fp32 accumulation order over K <= 2048,
through 50 + 16 layers measured 9.5e-6), token_num / L identical, ids identical wherever the oracle's top-1/top-2 margin
exceeds 1e-3.
"""
import numpy as np
import pytest

from aliparaformerasr_amd import weights as W
from oracle import frontend as fe
from oracle import glue
from oracle import model as om

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    T = max(f.shape[0] for f in feats)
    return fe.pad_sequence(feats).reshape(len(audio), T, 560)


def _ids_match(res, ref_logits, margin=1e-3):
    ids_ref = om.argmax_last(ref_logits)
    srt = np.sort(ref_logits, axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > margin
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(res.token_ids[safe], ids_ref[safe])
    return float((res.token_ids == ids_ref).mean())


@pytest.mark.parametrize("variant", ["loop", "cumsum"])
def test_fp32_mode_small_paraformer(variant):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=515)     # V not a multiple of 4
    cfg["cif_variant"] = variant
    w = W.synth_weights(cfg, seed=33)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=1)
    audio = [W.synth_audio(n, 5 + u) for u, n in enumerate((48000, 30000, 41000))]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").paraformer(speech)
    for res in (eng.forward_feats(speech, want_logits=True), eng.recognize(audio, want_logits=True)):
        np.testing.assert_array_equal(res.token_num, ref["token_num"])
        assert res.logits.shape == ref["logits"].shape
        err = np.abs(res.logits - ref["logits"]).max()
        assert err < TOL, err
        _ids_match(res, ref["logits"])
    ids_only = eng.recognize(audio)
    np.testing.assert_array_equal(ids_only.token_ids, res.token_ids)
    eng.close()


def test_fp32_mode_full_depth_is_closer_than_f16_mode():
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config()
    w = W.synth_weights(cfg, seed=42)
    cmvn = W.synth_cmvn()
    blob = W.pack_pfw(cfg, w)
    audio = [W.synth_audio(80000, u) for u in range(2)]
    speech = _speech(audio, cmvn)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32", fast=True).paraformer(speech)
    e32 = Engine(weights=blob, cmvn=cmvn, device=0, math_mode=1)
    r32 = e32.forward_feats(speech, want_logits=True)
    e32.close()
    e16 = Engine(weights=blob, cmvn=cmvn, device=0)
    r16 = e16.forward_feats(speech, want_logits=True)
    e16.close()
    np.testing.assert_array_equal(r32.token_num, ref["token_num"])
    err32 = np.abs(r32.logits - ref["logits"]).max()
    err16 = np.abs(r16.logits - ref["logits"]).max()
    print(f"full depth: |fp32 mode - oracle| = {err32:.2e}, |f16 mode - oracle| = {err16:.2e}")
    assert err32 < 5e-4, err32                       # Oracle(fast=True) uses torch kernels: a second summation order
    assert err32 * 10 < err16
    agree = _ids_match(r32, ref["logits"])
    assert agree > 0.99


def test_fp32_mode_sensevoice(sv_embed):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.sensevoice_small_config(enc_layers=3, tp_layers=2, vocab=403)
    w = W.synth_weights(cfg, seed=9)
    w["embed.weight"] = sv_embed.astype(np.float32)
    cmvn = W.synth_cmvn()
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=cmvn, device=0, math_mode=1)
    conf = fe.FrontendConf(dither=0.0)
    audio = [W.synth_audio(n, 70 + u) for u, n in enumerate((32000, 24000))]
    feats = [glue.sensevoice_prepend(fe.wav_frontend(a, conf, cmvn[0], cmvn[1]), sv_embed, use_itn=True) for a in audio]
    T = max(f.shape[0] for f in feats)
    speech = fe.pad_sequence(feats).reshape(2, T, 560)
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp32").sensevoice(speech)
    res = eng.forward_feats(speech, want_logits=True)
    assert res.logits.shape == ref["logits"].shape and res.L == T
    err = np.abs(res.logits - ref["logits"]).max()
    assert err < TOL, err
    _ids_match(res, ref["logits"])
    eng.close()


def test_fp32_mode_refuses_heads_it_does_not_cover():
    from aliparaformerasr_amd.engine import Engine
    from aliparaformerasr_amd._native import PfError
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=256, timestamp_head=True)
    w = W.synth_weights(cfg, seed=1)
    eng = Engine(weights=W.pack_pfw(cfg, w), cmvn=W.synth_cmvn(), device=0, math_mode=1)
    with pytest.raises(PfError) as ei:
        eng.recognize([W.synth_audio(16000, 1)])
    assert "math_mode" in str(ei.value)
    eng.close()

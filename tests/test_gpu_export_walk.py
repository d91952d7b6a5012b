"""GPU: a container converted from an EXPORTER-MADE ONNX graph (tests/funasr_like.py through torch.onnx.export)
run on the device through the C ABI against the oracle on the weights that went into the modules — the fp32 export on the f16
path, and the quantize_dynamic-style rewrite of the same graph (model.int8.onnx, the reference CLI's default,
Examples/OfflineAliParaformerAsrRecognizer.cs:17-22) in math_mode 2 against the int8 oracle on the container's stored bytes."""
import numpy as np
import pytest
import torch

import funasr_like as FL
from aliparaformerasr_amd import convert as cv, weights as W
from oracle import frontend as fe
from oracle import model as om

pytestmark = pytest.mark.gpu


def _speech(audio, cmvn):
    conf = fe.FrontendConf(dither=0.0)
    feats = [fe.wav_frontend(a, conf, cmvn[0], cmvn[1]) for a in audio]
    return fe.pad_sequence(feats).reshape(len(audio), -1, 560)


def test_exported_graph_to_container_to_device(tmp_path):
    from aliparaformerasr_amd.engine import Engine
    cfg = W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=515)
    w = W.synth_weights(cfg, seed=21)
    m = FL.Paraformer(cfg)
    FL.load_pfw_weights(m, cfg, w)
    blob = FL.export_onnx(m, (torch.randn(2, 12, 560),), ["speech"], ["logits", "alphas", "us_alphas"], {"speech": {0: "b", 1: "t"}})
    (tmp_path / "model.onnx").write_bytes(blob)
    assert cv.main([str(tmp_path / "model.onnx"), str(tmp_path / "model.pfw")]) == 0
    cmvn = W.synth_cmvn()
    audio = [W.synth_audio(n, 70 + u) for u, n in enumerate((40000, 31000))]
    speech = _speech(audio, cmvn)
    eng = Engine(weights_path=str(tmp_path / "model.pfw"), cmvn=cmvn, device=0)
    res = eng.recognize(audio, want_logits=True)
    eng.close()
    ref = om.Oracle(om.ModelConfig(**cfg), w, quant="fp16").paraformer(speech)          # the ORIGINAL weights
    np.testing.assert_array_equal(res.token_num, ref["token_num"])
    err = float(np.abs(res.logits - ref["logits"]).max())
    assert err < 5e-2, err
    srt = np.sort(ref["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 0.1
    np.testing.assert_array_equal(res.token_ids[safe], om.argmax_last(ref["logits"])[safe])
    print("exporter-made fp32 graph -> container -> device: max|dlogp| %.3e" % err)

    # ---- the int8 form of the same graph: stored bytes multiplied on the int8 matrix cores
    qblob, stored = FL.quantize_dynamic_rewrite(blob, exclude=("output",))
    (tmp_path / "model.int8.onnx").write_bytes(qblob)
    cfg8, w8 = cv.onnx_to_pfw(str(tmp_path / "model.int8.onnx"))
    assert sum(k.endswith(".weight_q") for k in w8) == len(stored)
    W.save_pfw(str(tmp_path / "model.int8.pfw"), cfg8, w8)
    eng = Engine(weights_path=str(tmp_path / "model.int8.pfw"), cmvn=cmvn, device=0, math_mode=2)
    r8 = eng.recognize(audio, want_logits=True)
    eng.close()
    ref8 = om.Oracle(om.ModelConfig(**cfg8), w8, quant="int8").paraformer(speech)      # the container's stored bytes + int8_exclude
    np.testing.assert_array_equal(r8.token_num, ref8["token_num"])
    e8 = np.abs(r8.logits - ref8["logits"])
    print("exporter-made int8 graph -> container -> device (math_mode 2): max|dlogp| %.3e mean %.2e" % (e8.max(), e8.mean()))
    assert e8.max() < 0.3 and e8.mean() < 3e-2
    srt = np.sort(ref8["logits"], axis=-1)
    safe = (srt[..., -1] - srt[..., -2]) > 0.3
    np.testing.assert_array_equal(r8.token_ids[safe], om.argmax_last(ref8["logits"])[safe])

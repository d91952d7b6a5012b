"""CPU: PFW1 container round trip and the synthetic generator's tensor inventory."""
import numpy as np

from aliparaformerasr_amd import weights as W


def test_pfw_roundtrip(tmp_path):
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=1, vocab=64)
    w = W.synth_weights(cfg, seed=1)
    blob = W.pack_pfw(cfg, w)
    assert blob[:4] == b"PFW1"
    cfg2, w2 = W.load_pfw(blob)
    assert cfg2 == cfg
    assert set(w2) == set(w)
    for k in w:
        np.testing.assert_array_equal(w[k], w2[k])
    p = tmp_path / "m.pfw"
    W.save_pfw(str(p), cfg, w)
    cfg3, w3 = W.load_pfw(str(p))
    assert cfg3 == cfg and set(w3) == set(w)


def test_inventory_shapes():
    cfg = W.paraformer_large_config(enc_layers=2, dec_layers=1, vocab=100)
    w = W.synth_weights(cfg, seed=2)
    assert w["encoder.layers.0.attn.qkv.weight"].shape == (1536, 560)
    assert w["encoder.layers.1.attn.qkv.weight"].shape == (1536, 512)
    assert w["encoder.layers.0.attn.fsmn.weight"].shape == (512, 11)
    assert w["predictor.conv.weight"].shape == (512, 512, 3)
    assert w["decoder.layers.0.src.kv.weight"].shape == (1024, 512)
    assert "decoder.layers.0.ffn.w2.bias" not in w
    assert w["decoder.output.weight"].shape == (100, 512)
    sv = W.synth_weights(W.sensevoice_small_config(enc_layers=1, tp_layers=1, vocab=50), seed=3)
    assert sv["ctc.weight"].shape == (50, 512) and sv["embed.weight"].shape == (16, 560)
    assert "encoder.tp_layers.0.ffn.w1.weight" in sv and "encoder.tp_norm.weight" in sv


def test_param_count_paraformer_large():
    # ~216 M active parameters
    cfg = W.paraformer_large_config()
    D, F, V = 512, 2048, 8404
    enc = (560 * 1536 + 1536) + 49 * (512 * 1536 + 1536) + 50 * (512 * 11 + 512 * 512 + 512 + 2 * D * F + F + D + 4 * D) + (560 - 512) * 2
    assert 150e6 < enc < 170e6
    assert cfg["enc_layers"] == 50 and cfg["dec_layers"] == 16 and cfg["vocab"] == V


def test_funasr_state_dict_mapping_round_trip():
    """convert.py: PFW inventory -> FunASR parameter names -> back (name/layout mapping only; not validated
    against a real checkpoint, see the module docstring).  Also: geometry inference, loud failures."""
    import numpy as np
    import pytest
    from aliparaformerasr_amd import convert as cv, weights as W
    for cfg in (W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=50),
                W.seaco_paraformer_config(enc_layers=2, dec_layers=1, vocab=40, seaco_layers=2),
                W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=30)):
        w = W.synth_weights(cfg, 1)
        nm = cv.name_map(cfg)
        assert set(nm) == set(w)                                    # every PFW tensor has a source name
        assert len(set(nm.values())) == len(nm)
        sd = {}
        for k, v in nm.items():
            a = w[k]
            sd[v] = a[:, None, :] if k.endswith("fsmn.weight") else a      # FunASR keeps depthwise conv weights [D,1,k]
        got = cv.state_dict_to_pfw(sd, cfg)
        assert all(np.array_equal(got[k], w[k]) for k in w)
        inf = cv.infer_config(sd, cfg["kind"])
        for key in ("enc_layers", "tp_layers", "dec_layers", "vocab", "kernel", "ffn", "feat_dim", "timestamp_head", "seaco_layers"):
            if cfg["kind"] == "sensevoicesmall" and key in ("timestamp_head", "seaco_layers"):
                continue
            if key == "seaco_layers" and not cfg["seaco"]:
                continue
            assert inf[key] == cfg[key], key
        bad = dict(sd)
        bad.pop(next(iter(nm.values())))
        with pytest.raises(KeyError):
            cv.state_dict_to_pfw(bad, cfg)
        bad = dict(sd)
        k0 = nm["encoder.layers.1.ffn.w1.weight"]
        bad[k0] = bad[k0][:-1]
        with pytest.raises(ValueError):
            cv.state_dict_to_pfw(bad, cfg)


def _synthetic_export(cfg, w, int8_names=(), stored=None):
    """Builds ONNX bytes shaped like a torch.onnx export of the FunASR modules: LayerNorm / Conv / bias tensors
    keep their parameter names (behind an 'encoder.model.'-style wrapper segment), Linear weights are anonymous
    transposed MatMul operands, LSTMs are ONNX LSTM nodes (gate order i,o,f,c), optionally int8-quantised."""
    import numpy as np
    from aliparaformerasr_amd import convert as cv, onnx_reader as R
    nm = cv.name_map(cfg)
    nodes, ini = [], {}
    anon = [0]
    H = 512
    def wrap(n):
        return n.replace("encoder.", "encoder.model.", 1) if n.startswith("encoder.") else n
    def to_iofc(a):
        blk = [a[k * H:(k + 1) * H] for k in range(4)]          # i f g o
        return np.concatenate([blk[0], blk[3], blk[1], blk[2]], 0)
    lstm_done = set()
    for k, fn in nm.items():
        a = w[k]
        if ".blstm." in fn or fn.startswith("bias_encoder."):
            grp = "blstm" if ".blstm." in fn else fn.split("_l")[-1].split("_")[0]
            if grp in lstm_done:
                continue
            lstm_done.add(grp)
            if grp == "blstm":
                keys = [("predictor.blstm.%s", ""), ("predictor.blstm.%s", "_reverse")]
                Wt = np.stack([to_iofc(w[p % "weight_ih" + s]) for p, s in keys])
                Rt = np.stack([to_iofc(w[p % "weight_hh" + s]) for p, s in keys])
                Bt = np.stack([np.concatenate([to_iofc(w[p % "bias_ih" + s]), to_iofc(w[p % "bias_hh" + s])]) for p, s in keys])
                attrs = {"direction": "bidirectional", "hidden_size": H}
            else:
                l = int(grp)
                Wt = to_iofc(w["seaco.lstm.l%d.weight_ih" % l])[None]
                Rt = to_iofc(w["seaco.lstm.l%d.weight_hh" % l])[None]
                Bt = np.concatenate([to_iofc(w["seaco.lstm.l%d.bias_ih" % l]), to_iofc(w["seaco.lstm.l%d.bias_hh" % l])])[None]
                attrs = {"hidden_size": H}
            base = "onnx::LSTM_%d" % anon[0]
            anon[0] += 3
            ini[base + "W"], ini[base + "R"], ini[base + "B"] = Wt, Rt, Bt
            nodes.append(("LSTM", "lstm_%s" % grp, ["x_%s" % grp, base + "W", base + "R", base + "B"], ["y_%s" % grp], attrs))
            continue
        is_linear_w = fn.endswith(".weight") and a.ndim == 2 and not fn.endswith(("norm.weight", "norm1.weight", "norm2.weight", "norm3.weight")) \
            and "fsmn_block" not in fn and fn not in ("bias_embed.weight", "embed.weight") and "cif_output" not in fn
        if is_linear_w:
            wn = "onnx::MatMul_%d" % anon[0]
            anon[0] += 1
            out = "mm_out_%d" % anon[0]
            has_bias = (fn[:-6] + "bias") in nm.values()
            src = "act_%d" % anon[0]
            if not has_bias:                                                       # w_2: fed by feed_forward.norm
                src = "ln_out_%d" % anon[0]
                pre = fn[: -len("w_2.weight")]
                nodes.append(("LayerNormalization", "ln%d" % anon[0], ["h_%d" % anon[0], wrap(pre + "norm.weight"), wrap(pre + "norm.bias")], [src], {}))
            if fn in int8_names and stored is not None:
                # quantize_dynamic(per_channel=True, weight_type=QUInt8): uint8 [K, N], scale / zero point per output channel
                from oracle.int8 import quantize_weight
                wq, ws, wz = quantize_weight(a)
                ini[wn + "_quantized"], ini[wn + "_scale"], ini[wn + "_zero_point"] = wq.T.astype(np.uint8), ws, wz.astype(np.uint8)
                w[k] = ((wq - wz[:, None]).astype(np.float32) * ws[:, None]).astype(np.float32)
                stored[k] = (wq.astype(np.uint8), wz.astype(np.uint8), ws)
            if fn in int8_names and stored is None:
                scale = np.float32(np.abs(a).max() / 127.0)
                qv = np.clip(np.round(a.T / scale), -127, 127).astype(np.int8)
                ini[wn + "_quantized"], ini[wn + "_scale"], ini[wn + "_zero_point"] = qv, np.asarray(scale, np.float32), np.asarray(0, np.int8)
                w[k] = (qv.astype(np.float32) * scale).T.copy()                  # what ingestion must return
            if fn in int8_names:
                nodes.append(("DynamicQuantizeLinear", "dq%d" % anon[0], [src], ["aq%d" % anon[0], "as%d" % anon[0], "az%d" % anon[0]], {}))
                nodes.append(("MatMulInteger", "mmi%d" % anon[0], ["aq%d" % anon[0], wn + "_quantized", "az%d" % anon[0], wn + "_zero_point"], ["mi%d" % anon[0]], {}))
                nodes.append(("Cast", "c%d" % anon[0], ["mi%d" % anon[0]], ["mc%d" % anon[0]], {"to": 1}))
                nodes.append(("Mul", "m%d" % anon[0], ["mc%d" % anon[0], "as%d" % anon[0]], [out], {}))
            else:
                ini[wn] = np.ascontiguousarray(a.T)
                nodes.append(("MatMul", "mm%d" % anon[0], [src, wn], [out], {}))
            if has_bias:
                nodes.append(("Add", "add%d" % anon[0], [out, wrap(fn[:-6] + "bias")], ["y_%d" % anon[0]], {}))
        else:
            a2 = a[:, None, :] if k.endswith("fsmn.weight") else a
            ini["embedding.weight" if fn == "bias_embed.weight" else wrap(fn)] = np.ascontiguousarray(a2, np.float32)
    return R.dump(nodes, ini, ["speech"], ["logits"])


def test_onnx_ingestion_round_trip_including_int8():
    """convert.onnx_to_pfw on synthetic export-shaped graphs (anonymous transposed MatMul weights named through
    their bias / preceding LayerNorm, ONNX LSTM gate order, quantize_dynamic naming, wrapper name segments)."""
    import numpy as np
    from aliparaformerasr_amd import convert as cv, onnx_reader as R, weights as W
    for cfg, q8 in ((W.paraformer_large_config(enc_layers=2, dec_layers=2, vocab=48, timestamp_head=True), ()),
                    (W.seaco_paraformer_config(enc_layers=2, dec_layers=1, vocab=40, seaco_layers=2),
                     ("encoder.encoders.0.feed_forward.w_1.weight", "decoder.decoders.0.feed_forward.w_2.weight", "decoder.output_layer.weight")),
                    (W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=30), ("ctc.ctc_lo.weight",))):
        w = W.synth_weights(cfg, 2)
        blob = _synthetic_export(cfg, w, set(q8))
        g = R.load(blob)
        sd = cv.onnx_to_state_dict(g, set(cv.name_map(cfg).values()))
        cfg2 = cv.infer_config(sd, cfg["kind"])
        got = cv.state_dict_to_pfw(sd, cfg2)
        qk = {k for k in got if k.endswith((".weight_q", ".weight_zp", ".weight_scale"))}
        assert set(got) - qk == set(w) and len(qk) == 3 * len(q8)
        for k in w:
            assert np.array_equal(got[k], w[k]), k
        for k in qk:                                  # a signed per-tensor export: bytes and zero point shifted by 128 together
            if k.endswith("_q"):
                stem = k[:-2]
                assert got[k].dtype == np.uint8 and got[k].shape == w[stem].shape
                deq = (got[k].astype(np.float32) - got[stem + "_zp"].astype(np.float32)[:, None]) * got[stem + "_scale"][:, None]
                assert np.array_equal(deq.astype(np.float32), w[stem]), k


def test_int8_export_bytes_travel_through_the_container():
    """An int8 export's stored operands (uint8 weights per output channel, quantize_dynamic's layout [K, N]) arrive in
    the container byte for byte (`<linear>.weight_q` [N, K], `.weight_zp`, `.weight_scale`) beside their de-quantised
    float image, survive pack / load, and the int8 oracle multiplies THEM (math_mode 2 does the same on the device,
    tests/test_gpu_int8.py)."""
    import numpy as np
    from aliparaformerasr_amd import convert as cv, onnx_reader as R, weights as W
    from oracle.int8 import QuantizedLinears, qlinear
    cfg = W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=30)
    w = W.synth_weights(cfg, 5)
    names = cv.name_map(cfg)
    q8 = {v for k, v in names.items() if k.endswith(".weight") and w[k].ndim == 2 and (".attn.qkv" in k or ".ffn." in k or k == "ctc.weight")}
    stored = {}
    blob = _synthetic_export(cfg, w, q8, stored)
    assert len(stored) == len(q8) == 3 * 3 + 1
    sd = cv.onnx_to_state_dict(R.load(blob), set(names.values()))
    got = cv.state_dict_to_pfw(sd, cv.infer_config(sd, cfg["kind"]))
    for k, (wq, wz, ws) in stored.items():
        assert np.array_equal(got[k + "_q"], wq) and got[k + "_q"].dtype == np.uint8, k
        assert np.array_equal(got[k + "_zp"], wz) and np.array_equal(got[k + "_scale"], ws), k
        assert np.array_equal(got[k], w[k]), k
    cfg2, back = W.load_pfw(W.pack_pfw(cfg, got))
    assert set(back) == set(got)
    for k in got:
        assert back[k].dtype == got[k].dtype and np.array_equal(back[k], got[k]), k
    # the oracle's quantised Linear uses the stored bytes, not a re-quantisation of the float image
    k = "ctc.weight"
    tampered = dict(back)
    tampered[k + "_q"] = (back[k + "_q"] ^ 1).astype(np.uint8)
    x = np.random.default_rng(0).standard_normal((5, back[k].shape[1])).astype(np.float32)
    y0 = QuantizedLinears(back)(x, "ctc", True)
    y1 = QuantizedLinears(tampered)(x, "ctc", True)
    want = qlinear(x, tampered[k + "_q"].astype(np.int32), back[k + "_scale"], back[k + "_zp"].astype(np.int32), back["ctc.bias"])
    assert np.array_equal(y1, want) and not np.array_equal(y0, y1)

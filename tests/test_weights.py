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

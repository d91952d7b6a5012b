"""CPU: convert.onnx_to_pfw on graphs made by a REAL exporter — the TorchScript ONNX exporter run on torch
modules with FunASR's parameter names and structure (tests/funasr_like.py) — and on a quantize_dynamic-style rewrite of
those graphs.  Until round 6 the walk had only seen graphs hand-built by tests/test_weights.py.  Reference call site:
AliParaformerAsr.Examples/OfflineAliParaformerAsrRecognizer.cs:17-22 (model.int8.onnx / model.onnx / model_eb.*.onnx)."""
import numpy as np
import pytest
import torch

import funasr_like as FL
from aliparaformerasr_amd import convert as cv, onnx_reader as R, weights as W


def _export(cfg, w, tmp_path=None):
    kind = cfg["kind"]
    if kind == "sensevoicesmall":
        m = FL.SenseVoiceSmall(cfg)
        FL.load_pfw_weights(m, cfg, w)
        blob = FL.export_onnx(m, (torch.randn(2, 9, cfg["feat_dim"]), torch.zeros(2, 4, dtype=torch.long)), ["speech", "prompt"], ["logits"],
                              {"speech": {0: "b", 1: "t"}})
        return blob, None
    m = FL.Paraformer(cfg)
    FL.load_pfw_weights(m, cfg, w)
    args = (torch.randn(2, 12, cfg["feat_dim"]),) + ((torch.randn(2, 7, cfg["d_model"]),) if cfg.get("seaco") else ())
    blob = FL.export_onnx(m, args, ["speech"] + (["bias_embed"] if cfg.get("seaco") else []),
                          ["logits", "alphas", "us_alphas"] + (["hot"] if cfg.get("seaco") else []), {"speech": {0: "b", 1: "t"}})
    eb = None
    if cfg.get("seaco"):
        e = FL.SeacoEmbedder(cfg)
        FL.load_pfw_weights(e, cfg, w)
        eb = FL.export_onnx(e, (torch.zeros(3, 10, dtype=torch.long),), ["hotword"], ["hw_embed"], {"hotword": {0: "n"}})
    return blob, eb


CFGS = [
    ("paraformer+ts", lambda: W.paraformer_large_config(enc_layers=3, dec_layers=2, vocab=48, timestamp_head=True)),
    ("seaco", lambda: W.seaco_paraformer_config(enc_layers=2, dec_layers=1, vocab=40, seaco_layers=2)),
    ("sensevoice", lambda: W.sensevoice_small_config(enc_layers=2, tp_layers=1, vocab=30)),
]


@pytest.mark.parametrize("name,mk", CFGS)
def test_converter_recovers_every_tensor_from_an_exporter_made_graph(name, mk, tmp_path):
    """fp32 export -> onnx_to_pfw (through FILES, as the CLI does) -> the geometry and EVERY tensor of the container equal
    the weights that went into the modules, bit for bit; the exported graph really has the properties the walk is for."""
    cfg = mk()
    w = W.synth_weights(cfg, 2)
    blob, eb = _export(cfg, w)
    g = R.load(blob)
    ops = [n.op_type for n in g.nodes]
    anon = [n.inputs[1] for n in g.nodes if n.op_type == "MatMul" and len(n.inputs) > 1 and n.inputs[1] in g.initializers]
    assert len(anon) >= 10 and all(a.startswith("onnx::MatMul") for a in anon), anon[:3]      # anonymous transposed Linear weights
    assert "LayerNormalization" in ops and "Conv" in ops
    if cfg.get("timestamp_head"):
        assert "LSTM" in ops and "ConvTranspose" in ops
    mp, ep = tmp_path / "model.onnx", tmp_path / "model_eb.onnx"
    mp.write_bytes(blob)
    if eb:
        ep.write_bytes(eb)
        assert "LSTM" in [n.op_type for n in R.load(eb).nodes]
    cfg2, got = cv.onnx_to_pfw(str(mp), str(ep) if eb else None, kind=cfg["kind"])
    diff = {k: (cfg[k], cfg2.get(k)) for k in cfg if cfg[k] != cfg2.get(k)}
    assert not diff, diff
    assert set(got) == set(w), set(got) ^ set(w)
    for k in w:
        assert got[k].dtype == w[k].dtype and np.array_equal(got[k], w[k]), k
    # ... and the CLI form writes a container the loader reads back
    out = tmp_path / "model.pfw"
    argv = [str(mp), str(out), "--kind", cfg["kind"]] + (["--eb", str(ep)] if eb else [])
    assert cv.main(argv) == 0
    cfg3, back = W.load_pfw(str(out))
    assert cfg3["kind"] == cfg["kind"] and all(np.array_equal(back[k], w[k]) for k in w)


@pytest.mark.parametrize("name,mk", CFGS[:2])
def test_converter_on_a_quantize_dynamic_rewrite_of_the_exported_graph(name, mk, tmp_path):
    """The int8 form of the same exporter-made graph (DynamicQuantizeLinear + MatMulInteger + Cast + Mul x 2 per MatMul, weights
    as <name>_quantized / _scale / _zero_point per output channel; the vocabulary projections excluded by node name as FunASR's
    export utility does): the container carries the STORED bytes of every quantised Linear ([N, K], zero points, scales)
    beside their de-quantised float image, the excluded Linears stay float and are listed in `int8_exclude`."""
    cfg = mk()
    w = W.synth_weights(cfg, 3)
    blob, eb = _export(cfg, w)
    g = R.load(blob)
    out_nodes = [n.name for n in g.nodes if n.op_type == "MatMul" and ("output_layer" in n.name)]
    assert out_nodes, "the exporter names nodes by module scope: /decoder/output_layer/MatMul"
    qblob, stored = FL.quantize_dynamic_rewrite(blob, exclude=("output",))
    gq = R.load(qblob)
    ops = [n.op_type for n in gq.nodes]
    assert ops.count("MatMulInteger") == len(stored) >= 10 and ops.count("DynamicQuantizeLinear") == ops.count("MatMulInteger")
    mp = tmp_path / "model.int8.onnx"
    mp.write_bytes(qblob)
    ep = None
    if eb:
        ep = tmp_path / "model_eb.int8.onnx"
        ep.write_bytes(eb)
    cfg2, got = cv.onnx_to_pfw(str(mp), str(ep) if ep else None, kind=cfg["kind"])
    qk = sorted(k for k in got if k.endswith(".weight_q"))
    assert len(qk) == len(stored)
    lin = {n.inputs[1]: n for n in g.nodes if n.op_type == "MatMul" and len(n.inputs) > 1 and n.inputs[1] in g.initializers}
    by_fp = {}
    for wname, (q, zp, sc) in stored.items():                     # key the stored triples by their de-quantised image
        by_fp[((q.astype(np.float32) - zp[:, None].astype(np.float32)) * sc[:, None]).astype(np.float32).tobytes()] = (q, zp, sc)
    for k in qk:
        stem = k[:-2]
        q, zp, sc = got[k], got[stem + "_zp"], got[stem + "_scale"]
        assert q.dtype == np.uint8 and q.shape == w[stem].shape
        deq = ((q.astype(np.float32) - zp[:, None].astype(np.float32)) * sc[:, None]).astype(np.float32)
        assert np.array_equal(got[stem], deq), stem                       # the float image IS the de-quantised stored operand
        assert deq.tobytes() in by_fp and np.array_equal(by_fp[deq.tobytes()][0], q), stem
        assert np.abs(deq - w[stem]).max() <= sc.max() * 0.5 + 1e-7, stem  # ... of the weight that went into the module
    # float Linears: exactly the excluded ones (+ the N = 1 predictor outputs are MatMuls too and get quantised by the rewrite)
    floats = [k for k in got if k.endswith(".weight") and got[k].ndim == 2 and k + "_q" not in got and
              (k[:-7] + ".bias" in got or k.endswith("ffn.w2.weight"))]
    assert any("output" in k for k in floats) and all(np.array_equal(got[k], w[k]) for k in floats)
    assert set(cfg2.get("int8_exclude", ())) >= {k[:-7] for k in floats if "output" in k}

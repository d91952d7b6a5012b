"""FunASR PyTorch checkpoint (state_dict) -> PFW1 container.

The reference consumes ONNX exports of FunASR models (README.EN.md:233); every export is produced from a FunASR
PyTorch checkpoint (`model.pt`) whose parameter names are fixed by the public FunASR module definitions
(SANMEncoder / CifPredictorV2|V3 / ParaformerSANMDecoder / SeacoParaformer / SenseVoiceSmall).  This module maps
those names onto the PFW tensor inventory of `aliparaformerasr_amd/weights.py`.  It is a NAME/LAYOUT mapping only
(no arithmetic); ONNX graph-walk ingestion incl. the int8 variants stays a "next" item.

NOT VALIDATED AGAINST A REAL CHECKPOINT: no FunASR file exists in the build image; the table below is restated
from the public FunASR sources and exercised by a round-trip test on synthetic tensors (tests/test_weights.py).
Every tensor is shape-checked against the target geometry, and unknown / missing names are reported, so a
mismatch with a real file fails loudly instead of producing a silently wrong model.

    python -m aliparaformerasr_amd.convert model.pt out.pfw [--kind paraformer|seacoparaformer|sensevoicesmall]
                                                           [--timestamp]
    python -m aliparaformerasr_amd.convert model.int8.onnx out.pfw [--eb model_eb.int8.onnx] [--kind ...]
        (ONNX graph walk, see onnx_to_state_dict; an int8 file's stored bytes are carried beside their float image)
"""
from __future__ import annotations

import re
import sys

import numpy as np

from . import weights as W

_ENC = {  # PFW suffix -> FunASR suffix inside an encoder block
    "norm1.weight": "norm1.weight", "norm1.bias": "norm1.bias", "norm2.weight": "norm2.weight", "norm2.bias": "norm2.bias",
    "attn.qkv.weight": "self_attn.linear_q_k_v.weight", "attn.qkv.bias": "self_attn.linear_q_k_v.bias",
    "attn.fsmn.weight": "self_attn.fsmn_block.weight",
    "attn.out.weight": "self_attn.linear_out.weight", "attn.out.bias": "self_attn.linear_out.bias",
    "ffn.w1.weight": "feed_forward.w_1.weight", "ffn.w1.bias": "feed_forward.w_1.bias",
    "ffn.w2.weight": "feed_forward.w_2.weight", "ffn.w2.bias": "feed_forward.w_2.bias",
}
_DEC = {
    "norm1.weight": "norm1.weight", "norm1.bias": "norm1.bias", "norm2.weight": "norm2.weight", "norm2.bias": "norm2.bias",
    "norm3.weight": "norm3.weight", "norm3.bias": "norm3.bias",
    "ffn.w1.weight": "feed_forward.w_1.weight", "ffn.w1.bias": "feed_forward.w_1.bias",
    "ffn.norm.weight": "feed_forward.norm.weight", "ffn.norm.bias": "feed_forward.norm.bias",
    "ffn.w2.weight": "feed_forward.w_2.weight",
    "fsmn.weight": "self_attn.fsmn_block.weight",
    "src.q.weight": "src_attn.linear_q.weight", "src.q.bias": "src_attn.linear_q.bias",
    "src.kv.weight": "src_attn.linear_k_v.weight", "src.kv.bias": "src_attn.linear_k_v.bias",
    "src.out.weight": "src_attn.linear_out.weight", "src.out.bias": "src_attn.linear_out.bias",
}
_DEC_FINAL = {k: v for k, v in _DEC.items() if k.startswith(("norm1", "ffn"))}


def name_map(cfg: dict) -> dict:
    """PFW tensor name -> FunASR state_dict key, for the geometry in cfg."""
    m = {}
    for i in range(cfg["enc_layers"]):
        src = "encoder.encoders0.0" if i == 0 else "encoder.encoders.%d" % (i - 1)
        for a, b in _ENC.items():
            m["encoder.layers.%d.%s" % (i, a)] = "%s.%s" % (src, b)
    m["encoder.after_norm.weight"], m["encoder.after_norm.bias"] = "encoder.after_norm.weight", "encoder.after_norm.bias"
    for i in range(cfg["tp_layers"]):
        for a, b in _ENC.items():
            m["encoder.tp_layers.%d.%s" % (i, a)] = "encoder.tp_encoders.%d.%s" % (i, b)
    if cfg["tp_layers"]:
        m["encoder.tp_norm.weight"], m["encoder.tp_norm.bias"] = "encoder.tp_norm.weight", "encoder.tp_norm.bias"
    if cfg["kind"] == "sensevoicesmall":
        m["ctc.weight"], m["ctc.bias"] = "ctc.ctc_lo.weight", "ctc.ctc_lo.bias"
        m["embed.weight"] = "embed.weight"
        return m
    m["predictor.conv.weight"], m["predictor.conv.bias"] = "predictor.cif_conv1d.weight", "predictor.cif_conv1d.bias"
    m["predictor.out.weight"], m["predictor.out.bias"] = "predictor.cif_output.weight", "predictor.cif_output.bias"
    if cfg.get("timestamp_head"):
        m["predictor.upsample.weight"], m["predictor.upsample.bias"] = "predictor.upsample_cnn.weight", "predictor.upsample_cnn.bias"
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                m["predictor.blstm.%s%s" % (nm, sfx)] = "predictor.blstm.%s_l0%s" % (nm, sfx)
        m["predictor.out2.weight"], m["predictor.out2.bias"] = "predictor.cif_output2.weight", "predictor.cif_output2.bias"
    for i in range(cfg["dec_layers"]):
        for a, b in _DEC.items():
            m["decoder.layers.%d.%s" % (i, a)] = "decoder.decoders.%d.%s" % (i, b)
    for a, b in _DEC_FINAL.items():
        m["decoder.final.%s" % a] = "decoder.decoders3.0.%s" % b
    m["decoder.after_norm.weight"], m["decoder.after_norm.bias"] = "decoder.after_norm.weight", "decoder.after_norm.bias"
    m["decoder.output.weight"], m["decoder.output.bias"] = "decoder.output_layer.weight", "decoder.output_layer.bias"
    if cfg.get("seaco"):
        m["seaco.embed.weight"] = "bias_embed.weight"
        for l in range(cfg["seaco_lstm_layers"]):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                m["seaco.lstm.l%d.%s" % (l, nm)] = "bias_encoder.%s_l%d" % (nm, l)
        for i in range(cfg["seaco_layers"]):
            for a, b in _DEC.items():
                m["seaco.decoder.layers.%d.%s" % (i, a)] = "seaco_decoder.decoders.%d.%s" % (i, b)
        for a, b in _DEC_FINAL.items():
            m["seaco.decoder.final.%s" % a] = "seaco_decoder.decoders3.0.%s" % b
        m["seaco.decoder.after_norm.weight"] = "seaco_decoder.after_norm.weight"
        m["seaco.decoder.after_norm.bias"] = "seaco_decoder.after_norm.bias"
        m["seaco.output.weight"], m["seaco.output.bias"] = "hotword_output_layer.weight", "hotword_output_layer.bias"
    return m


def _layout(pfw_name: str, a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, np.float32)
    if pfw_name.endswith("fsmn.weight") and a.ndim == 3:            # depthwise Conv1d [D,1,k] -> [D,k]
        a = a[:, 0, :]
    return np.ascontiguousarray(a)


def infer_config(sd: dict, kind: str = "paraformer", timestamp: bool | None = None) -> dict:
    """Geometry from the checkpoint itself (layer counts, vocabulary, kernel sizes)."""
    def count(pat):
        ids = {int(m.group(1)) for k in sd for m in [re.match(pat, k)] if m}
        return max(ids) + 1 if ids else 0
    kw = dict(kind=kind, enc_layers=1 + count(r"encoder\.encoders\.(\d+)\."), tp_layers=count(r"encoder\.tp_encoders\.(\d+)\."))
    kw["kernel"] = int(np.asarray(sd["encoder.encoders0.0.self_attn.fsmn_block.weight"]).shape[-1])
    kw["ffn"] = int(np.asarray(sd["encoder.encoders0.0.feed_forward.w_1.weight"]).shape[0])
    kw["feat_dim"] = int(np.asarray(sd["encoder.encoders0.0.self_attn.linear_q_k_v.weight"]).shape[1])
    if kind == "sensevoicesmall":
        kw.update(dec_layers=0, vocab=int(np.asarray(sd["ctc.ctc_lo.weight"]).shape[0]))
        return W.make_config(**kw)
    kw["dec_layers"] = count(r"decoder\.decoders\.(\d+)\.")
    kw["vocab"] = int(np.asarray(sd["decoder.output_layer.weight"]).shape[0])
    kw["timestamp_head"] = ("predictor.upsample_cnn.weight" in sd) if timestamp is None else bool(timestamp)
    if kind == "seacoparaformer":
        kw.update(seaco=True, seaco_layers=count(r"seaco_decoder\.decoders\.(\d+)\."),
                  seaco_ffn=int(np.asarray(sd["seaco_decoder.decoders.0.feed_forward.w_1.weight"]).shape[0]),
                  seaco_kernel=int(np.asarray(sd["seaco_decoder.decoders.0.self_attn.fsmn_block.weight"]).shape[-1]),
                  seaco_lstm_layers=count(r"bias_encoder\.weight_ih_l(\d+)$"))
    return W.make_config(**kw)


def state_dict_to_pfw(sd: dict, cfg: dict) -> dict:
    """-> PFW weight dict; raises KeyError listing every missing checkpoint key, ValueError on a shape clash
    with the synthetic inventory of the same geometry (which is what the engine will bind)."""
    nm = name_map(cfg)
    missing = [v for v in nm.values() if v not in sd]
    if missing:
        raise KeyError("checkpoint lacks %d expected tensors, e.g. %s" % (len(missing), missing[:5]))
    out = {k: _layout(k, sd[v]) for k, v in nm.items()}
    probe = dict(cfg, vocab=min(cfg["vocab"], 64))                  # shapes only: keep the reference inventory small
    ref = W.synth_weights(probe, 0)
    for k, a in out.items():
        want = list(ref[k].shape)
        got = list(a.shape)
        if k in ("decoder.output.weight", "decoder.output.bias", "ctc.weight", "ctc.bias", "seaco.output.weight",
                 "seaco.output.bias", "seaco.embed.weight"):
            want[0] = got[0]                                        # vocabulary-sized
        if want != got:
            raise ValueError("%s: shape %s, expected %s" % (k, got, want))
    for k, v in nm.items():                                         # an int8 export's stored bytes travel beside the float image
        if v + "_q" in sd and v + "_zp" in sd and v + "_scale" in sd and np.asarray(sd[v + "_q"]).shape == out[k].shape:
            out[k + "_q"] = np.ascontiguousarray(sd[v + "_q"], np.uint8)
            out[k + "_zp"] = np.ascontiguousarray(sd[v + "_zp"], np.uint8)
            out[k + "_scale"] = np.ascontiguousarray(sd[v + "_scale"], np.float32)
    return out


# ------------------------------------------------------------------ ONNX ingestion ---------
_LSTM_IOFC_TO_IFGO = (0, 2, 3, 1)      # ONNX LSTM gate blocks are i,o,f,c; PyTorch (and the engine) use i,f,g(=c),o


def _reorder_gates(a, H):
    blocks = [a[k * H:(k + 1) * H] for k in range(4)]
    return np.concatenate([blocks[k] for k in _LSTM_IOFC_TO_IFGO], axis=0)


_Q_SUFFIXES = ("_q", "_zp", "_scale")         # `<linear>.weight` + suffix: the stored bytes of a quantised Linear


def onnx_to_state_dict(graph, expected_names) -> dict:
    """Initializers of a FunASR ONNX export -> {FunASR parameter name: float32 array}.

    * `X_quantized` + `X_scale` + `X_zero_point` (onnxruntime quantize_dynamic naming) are de-quantised to X (what
      math modes 0 / 1 multiply) AND carried as stored: `<linear>.weight_q` (uint8 [N, K]; a signed export's bytes are
      shifted by 128 together with the zero point, which leaves q - zp unchanged), `<linear>.weight_zp` (uint8 [N]),
      `<linear>.weight_scale` (float32 [N]; a per-tensor scale is repeated) — what math_mode 2 multiplies.
    * torch.onnx exports nn.Linear on 3-D inputs as MatMul(x, W^T) with an anonymous initializer followed by
      Add(bias): the weight is named after the bias it feeds; the bias-free decoder `feed_forward.w_2` is named
      after the `feed_forward.norm` LayerNorm that produces its input.
    * LSTM nodes carry W/R/B in ONNX layout [dirs, 4H, ...] with gate order i,o,f,c: split per direction, gates
      re-ordered; bidirectional -> predictor.blstm (BiCIF), unidirectional -> bias_encoder layers in graph order.
    * names are matched on suffixes against `expected_names` (export wrappers prepend / insert module names).
    Unvalidated against a real export (none exists in the build image); everything unmatched is ignored here and
    reported by state_dict_to_pfw as missing."""
    ini = dict(graph.initializers)
    flt = {}
    stored = {}                      # base -> (q uint8 [K, N], zp uint8 [N], scale float32 [N]) of a 2-D quantised MatMul operand
    for name, arr in ini.items():
        if name.endswith("_quantized") and (name[:-10] + "_scale") in ini:
            base = name[:-10]
            scale = np.asarray(ini[base + "_scale"], np.float32)
            zpi = np.asarray(ini.get(base + "_zero_point", 0))
            zp = zpi.astype(np.float32)
            flt[base] = ((arr.astype(np.float32) - zp) * scale).astype(np.float32)
            if arr.ndim == 2 and arr.dtype in (np.uint8, np.int8) and scale.size in (1, arr.shape[1]) and zpi.size in (1, arr.shape[1]):
                shift = 128 if arr.dtype == np.int8 else 0
                stored[base] = ((arr.astype(np.int32) + shift).astype(np.uint8),
                                np.broadcast_to((zpi.astype(np.int32) + shift).astype(np.uint8).reshape(-1), (arr.shape[1],)).copy(),
                                np.broadcast_to(scale.reshape(-1), (arr.shape[1],)).astype(np.float32).copy())
        elif arr.dtype in (np.float32, np.float16, np.float64) and not name.endswith(("_scale", "_zero_point")):
            flt[name] = arr.astype(np.float32)

    def base_of(n):
        return n[:-10] if n.endswith("_quantized") else n

    sd = dict(flt)
    through = ("Cast", "Mul", "Add", "Reshape", "DynamicQuantizeLinear", "Identity")
    for node in graph.nodes:
        if node.op_type in ("MatMul", "MatMulInteger") and len(node.inputs) >= 2 and base_of(node.inputs[1]) in flt:
            wt = flt[base_of(node.inputs[1])]
            target = None
            frontier, depth = [node.outputs[0]], 0
            while frontier and depth < 5 and target is None:              # forward: the bias Add
                nxt = []
                for val in frontier:
                    for c in graph.consumers(val):
                        if c.op_type == "Add":
                            for i in c.inputs:
                                if i in flt and i.endswith(".bias") and flt[i].ndim == 1 and flt[i].shape[0] == wt.shape[-1]:
                                    target = i[:-4] + "weight"
                        if c.op_type in through:
                            nxt += c.outputs
                frontier, depth = nxt, depth + 1
            if target is None:                                             # backward: bias-free w_2 after feed_forward.norm
                val, depth = node.inputs[0], 0
                while val and depth < 8 and target is None:
                    pr = graph.producer(val)
                    if pr is None:
                        break
                    for i in pr.inputs:
                        if i in flt and ".feed_forward.norm." in i:
                            target = i.split(".feed_forward.norm.")[0] + ".feed_forward.w_2.weight"
                    val, depth = (pr.inputs[0] if pr.inputs else None), depth + 1
            if target is not None and target not in sd:
                sd[target] = np.ascontiguousarray(wt.T)
                if base_of(node.inputs[1]) in stored:
                    q, zp, sc = stored[base_of(node.inputs[1])]
                    sd[target + "_q"], sd[target + "_zp"], sd[target + "_scale"] = np.ascontiguousarray(q.T), zp, sc
    uni = 0
    for node in graph.nodes:
        if node.op_type != "LSTM" or len(node.inputs) < 3:
            continue
        Wt, Rt = flt.get(base_of(node.inputs[1])), flt.get(base_of(node.inputs[2]))
        Bt = flt.get(base_of(node.inputs[3])) if len(node.inputs) > 3 and node.inputs[3] else None
        if Wt is None or Rt is None:
            continue
        H = Rt.shape[-1]
        bidir = str(node.attrs.get("direction", "forward")) == "bidirectional" or Wt.shape[0] == 2
        for d in range(Wt.shape[0]):
            if bidir:
                pre, sfx = "predictor.blstm.", "_l0" + ("_reverse" if d == 1 else "")
            else:
                pre, sfx = "bias_encoder.", "_l%d" % uni
            sd[pre + "weight_ih" + sfx] = _reorder_gates(Wt[d], H)
            sd[pre + "weight_hh" + sfx] = _reorder_gates(Rt[d], H)
            if Bt is not None:
                sd[pre + "bias_ih" + sfx] = _reorder_gates(Bt[d][:4 * H], H)
                sd[pre + "bias_hh" + sfx] = _reorder_gates(Bt[d][4 * H:], H)
        if not bidir:
            uni += 1
    # suffix normalisation onto the expected FunASR names
    out = {}
    aliases = {"embedding.weight": "bias_embed.weight", "embed.weight": "embed.weight"}
    exp = sorted(expected_names, key=len, reverse=True)
    for k, v in sd.items():
        sfx = next((x for x in _Q_SUFFIXES if k.endswith(".weight" + x)), "")
        k0 = k[:len(k) - len(sfx)]
        cands = {k0, k0.replace(".model.", "."), aliases.get(k0, k0)}
        for e in exp:
            if any(c == e or c.endswith("." + e) for c in cands):
                out.setdefault(e + sfx, v)
                break
    return out


def onnx_to_pfw(model_path, eb_path=None, kind="paraformer", timestamp=None):
    """model.onnx / model.int8.onnx (+ model_eb*.onnx for SeACo) -> (cfg, PFW weights)."""
    from . import onnx_reader as R
    graphs = [R.load(model_path)] + ([R.load(eb_path)] if eb_path else [])
    # every name any supported geometry could ask for (layer indices bounded generously; unmatched ones are simply absent)
    probe = W.make_config(kind=kind, enc_layers=64, tp_layers=32 if kind == "sensevoicesmall" else 0, dec_layers=32,
                          timestamp_head=True, seaco=(kind == "seacoparaformer"), seaco_layers=8, seaco_lstm_layers=4)
    expected = set(name_map(probe).values())
    sd = {}
    for g in graphs:
        sd.update(onnx_to_state_dict(g, expected))
    cfg = infer_config(sd, kind, timestamp)
    wts = state_dict_to_pfw(sd, cfg)
    # an int8 export: record which Linears the export left in float (FunASR passes nodes_to_exclude to quantize_dynamic).
    # The engine and the oracle decide by the stored bytes themselves; the key documents the set in the container.
    if any(k.endswith(".weight_q") for k in wts):
        lin = [k[:-7] for k, v in wts.items() if k.endswith(".weight") and np.asarray(v).ndim == 2 and k[:-7] + ".bias" in wts
               or k.endswith(".ffn.w2.weight")]
        cfg["int8_exclude"] = tuple(sorted(n for n in lin if n + ".weight_q" not in wts))
    return cfg, wts


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        print(__doc__)
        return 2
    kind, ts = "paraformer", None
    if "--kind" in argv:
        kind = argv[argv.index("--kind") + 1]
    if "--timestamp" in argv:
        ts = True
    if argv[0].endswith(".onnx"):
        eb = argv[argv.index("--eb") + 1] if "--eb" in argv else None
        cfg, wts = onnx_to_pfw(argv[0], eb, kind, ts)
        W.save_pfw(argv[1], cfg, wts)
        print("wrote %s (%s, %d tensors, from ONNX)" % (argv[1], cfg["kind"], len(wts)))
        return 0
    import torch
    sd = torch.load(argv[0], map_location="cpu")
    sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd
    sd = {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "numpy")}
    cfg = infer_config(sd, kind, ts)
    W.save_pfw(argv[1], cfg, state_dict_to_pfw(sd, cfg))
    print("wrote %s (%s, %d tensors)" % (argv[1], cfg["kind"], len(name_map(cfg))))
    return 0


if __name__ == "__main__":
    sys.exit(main())

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
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        print(__doc__)
        return 2
    import torch
    kind, ts = "paraformer", None
    if "--kind" in argv:
        kind = argv[argv.index("--kind") + 1]
    if "--timestamp" in argv:
        ts = True
    sd = torch.load(argv[0], map_location="cpu")
    sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd
    sd = {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "numpy")}
    cfg = infer_config(sd, kind, ts)
    W.save_pfw(argv[1], cfg, state_dict_to_pfw(sd, cfg))
    print("wrote %s (%s, %d tensors)" % (argv[1], cfg["kind"], len(name_map(cfg))))
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Test infrastructure: torch modules with the PUBLIC FunASR parameter names and module structure (SANMEncoder /
CifPredictorV3 / ParaformerSANMDecoder / SenseVoice CTC head / SeACo hot-word embedder), small enough to export with
the TorchScript ONNX exporter in the build image.

Why: `aliparaformerasr_amd/convert.py::onnx_to_pfw` had only ever seen graphs hand-built by
tests/test_weights.py.  What `torch.onnx.export` makes of these modules is the real thing for every property the walk
depends on — Linear on 3-D inputs as MatMul against an ANONYMOUS transposed initialiser followed by Add(<named bias>), the
bias-free `feed_forward.w_2` behind `feed_forward.norm`, fused LayerNormalization (opset 17), Conv / ConvTranspose weights
under their parameter names, ONNX LSTM nodes with [dirs, 4H, .] operands in i,o,f,c gate order, Gather on an embedding
table, the exporter's node order and value names.  The arithmetic of the forward pass is a plausible SAN-M (it only
has to trace); parity of the ARITHMETIC is the oracle's job, not this file's.

`export_onnx` needs no `onnx` package: the exporter's C++ serialiser writes the ModelProto; the one Python step that
imports `onnx` (merging onnxscript functions, of which there are none here) is bypassed.
"""
import io
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.w_1, self.w_2 = nn.Linear(d, f), nn.Linear(f, d)

    def forward(self, x):
        return self.w_2(torch.relu(self.w_1(x)))


class PositionwiseFeedForwardDecoderSANM(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.w_1, self.w_2, self.norm = nn.Linear(d, f), nn.Linear(f, d, bias=False), nn.LayerNorm(f)

    def forward(self, x):
        return self.w_2(self.norm(torch.relu(self.w_1(x))))


def _fsmn(conv, v, kernel):
    left = (kernel - 1) // 2
    y = conv(F.pad(v.transpose(1, 2), (left, kernel - 1 - left)))
    return y.transpose(1, 2) + v


class MultiHeadedAttentionSANM(nn.Module):
    def __init__(self, h, d_in, d, kernel):
        super().__init__()
        self.h, self.d, self.kernel = h, d, kernel
        self.linear_q_k_v, self.linear_out = nn.Linear(d_in, 3 * d), nn.Linear(d, d)
        self.fsmn_block = nn.Conv1d(d, d, kernel, groups=d, bias=False)

    def forward(self, x):
        q, k, v = torch.split(self.linear_q_k_v(x), self.d, dim=-1)
        B, T, dk = x.shape[0], x.shape[1], self.d // self.h
        qh = q.reshape(B, T, self.h, dk).transpose(1, 2) * dk ** -0.5
        kh = k.reshape(B, T, self.h, dk).transpose(1, 2)
        vh = v.reshape(B, T, self.h, dk).transpose(1, 2)
        ctx = torch.softmax(qh @ kh.transpose(-2, -1), dim=-1) @ vh
        return self.linear_out(ctx.transpose(1, 2).reshape(B, T, self.d)) + _fsmn(self.fsmn_block, v, self.kernel)


class EncoderLayerSANM(nn.Module):
    def __init__(self, d_in, d, f, h, kernel):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(d_in), nn.LayerNorm(d)
        self.self_attn = MultiHeadedAttentionSANM(h, d_in, d, kernel)
        self.feed_forward = PositionwiseFeedForward(d, f)
        self.same = d_in == d

    def forward(self, x):
        a = self.self_attn(self.norm1(x))
        x = x + a if self.same else a
        return x + self.feed_forward(self.norm2(x))


class SANMEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d, f, h, k = cfg["d_model"], cfg["ffn"], cfg["heads"], cfg["kernel"]
        self.encoders0 = nn.ModuleList([EncoderLayerSANM(cfg["feat_dim"], d, f, h, k)])
        self.encoders = nn.ModuleList([EncoderLayerSANM(d, d, f, h, k) for _ in range(cfg["enc_layers"] - 1)])
        self.after_norm = nn.LayerNorm(d)
        self.tp_encoders = nn.ModuleList([EncoderLayerSANM(d, d, f, h, k) for _ in range(cfg["tp_layers"])])
        if cfg["tp_layers"]:
            self.tp_norm = nn.LayerNorm(d)
        self.scale = float(d) ** 0.5

    def forward(self, x):
        x = x * self.scale
        for layer in list(self.encoders0) + list(self.encoders):
            x = layer(x)
        x = self.after_norm(x)
        if len(self.tp_encoders):
            for layer in self.tp_encoders:
                x = layer(x)
            x = self.tp_norm(x)
        return x


class CifPredictorV3(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["d_model"]
        self.cif_conv1d = nn.Conv1d(d, d, cfg["cif_l_order"] + cfg["cif_r_order"] + 1, padding=cfg["cif_l_order"])
        self.cif_output = nn.Linear(d, 1)
        self.ts = bool(cfg.get("timestamp_head"))
        if self.ts:
            self.upsample_cnn = nn.ConvTranspose1d(d, d, 3, 3)
            self.blstm = nn.LSTM(d, d, 1, batch_first=True, bidirectional=True)
            self.cif_output2 = nn.Linear(2 * d, 1)

    def forward(self, h):
        y = torch.relu(self.cif_conv1d(h.transpose(1, 2)).transpose(1, 2) + h)
        alphas = torch.sigmoid(self.cif_output(y))
        if not self.ts:
            return alphas, alphas
        up = torch.relu(self.upsample_cnn(h.transpose(1, 2)).transpose(1, 2))
        z, _ = self.blstm(up)
        return alphas, torch.sigmoid(self.cif_output2(z))


class MultiHeadedAttentionSANMDecoder(nn.Module):
    def __init__(self, d, kernel):
        super().__init__()
        self.kernel = kernel
        self.fsmn_block = nn.Conv1d(d, d, kernel, groups=d, bias=False)

    def forward(self, x):
        return _fsmn(self.fsmn_block, x, self.kernel)


class MultiHeadedAttentionCrossAtt(nn.Module):
    def __init__(self, h, d):
        super().__init__()
        self.h, self.d = h, d
        self.linear_q, self.linear_k_v, self.linear_out = nn.Linear(d, d), nn.Linear(d, 2 * d), nn.Linear(d, d)

    def forward(self, x, mem):
        B, L, T, dk = x.shape[0], x.shape[1], mem.shape[1], self.d // self.h
        q = self.linear_q(x).reshape(B, L, self.h, dk).transpose(1, 2) * dk ** -0.5
        k, v = torch.split(self.linear_k_v(mem), self.d, dim=-1)
        kh, vh = k.reshape(B, T, self.h, dk).transpose(1, 2), v.reshape(B, T, self.h, dk).transpose(1, 2)
        ctx = torch.softmax(q @ kh.transpose(-2, -1), dim=-1) @ vh
        return self.linear_out(ctx.transpose(1, 2).reshape(B, L, self.d))


class DecoderLayerSANM(nn.Module):
    def __init__(self, d, f, h, kernel, full=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(d)
        self.feed_forward = PositionwiseFeedForwardDecoderSANM(d, f)
        self.full = full
        if full:
            self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d)
            self.self_attn = MultiHeadedAttentionSANMDecoder(d, kernel)
            self.src_attn = MultiHeadedAttentionCrossAtt(h, d)

    def forward(self, x, mem):
        t = self.feed_forward(self.norm1(x))
        if not self.full:
            return t
        x = x + self.self_attn(self.norm2(t))
        return x + self.src_attn(self.norm3(x), mem)


class ParaformerSANMDecoder(nn.Module):
    def __init__(self, d, f, h, kernel, layers, vocab=None):
        super().__init__()
        self.decoders = nn.ModuleList([DecoderLayerSANM(d, f, h, kernel) for _ in range(layers)])
        self.decoders3 = nn.ModuleList([DecoderLayerSANM(d, f, h, kernel, full=False)])
        self.after_norm = nn.LayerNorm(d)
        if vocab:
            self.output_layer = nn.Linear(d, vocab)

    def forward(self, x, mem):
        for layer in self.decoders:
            x = layer(x, mem)
        x = self.after_norm(self.decoders3[0](x, mem))
        return self.output_layer(x) if hasattr(self, "output_layer") else x


class Paraformer(nn.Module):
    """model.onnx: speech [B, T, 560] -> (log-probs [B, L, V], alphas, us_alphas); with `seaco` also the bias decoder and
    the hot-word output layer fed with `bias_embed` [B, N, 512] (the third graph input of the SeACo export)."""

    def __init__(self, cfg, L=5):
        super().__init__()
        d, h = cfg["d_model"], cfg["heads"]
        self.L = L
        self.encoder = SANMEncoder(cfg)
        self.predictor = CifPredictorV3(cfg)
        self.decoder = ParaformerSANMDecoder(d, cfg["ffn"], h, cfg["kernel"], cfg["dec_layers"], cfg["vocab"])
        self.seaco = bool(cfg.get("seaco"))
        if self.seaco:
            self.seaco_decoder = ParaformerSANMDecoder(d, cfg["seaco_ffn"], h, cfg["seaco_kernel"], cfg["seaco_layers"])
            self.hotword_output_layer = nn.Linear(d, cfg["vocab"])

    def forward(self, speech, bias_embed=None):
        mem = self.encoder(speech)
        alphas, us = self.predictor(mem)
        emb = mem[:, : self.L] * alphas[:, : self.L]              # stand-in for the integrate-and-fire loop (no parameters there)
        logp = torch.log_softmax(self.decoder(emb, mem), dim=-1)
        if not self.seaco:
            return logp, alphas, us
        hot = self.hotword_output_layer(self.seaco_decoder(emb, bias_embed))
        return logp, alphas, us, torch.log_softmax(hot, dim=-1)


class SenseVoiceSmall(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = SANMEncoder(cfg)
        self.ctc = nn.Module()
        self.ctc.ctc_lo = nn.Linear(cfg["d_model"], cfg["vocab"])
        self.embed = nn.Embedding(16, cfg["feat_dim"])

    def forward(self, speech, prompt_ids):
        x = torch.cat([self.embed(prompt_ids), speech], dim=1)
        return torch.log_softmax(self.ctc.ctc_lo(self.encoder(x)), dim=-1)


class SeacoEmbedder(nn.Module):
    """model_eb.onnx: hot-word ids [N, 10] -> [10, N, 512] (EmbedSeacoModel.cs:70-123)."""

    def __init__(self, cfg):
        super().__init__()
        self.bias_embed = nn.Embedding(cfg["vocab"], cfg["d_model"])
        self.bias_encoder = nn.LSTM(cfg["d_model"], cfg["d_model"], cfg["seaco_lstm_layers"], batch_first=True)

    def forward(self, ids):
        y, _ = self.bias_encoder(self.bias_embed(ids))
        return y.transpose(0, 1)


def load_pfw_weights(module, cfg, w, keys=None):
    """PFW weight dict -> the module's parameters through convert.name_map (the FunASR key of every PFW tensor)."""
    from aliparaformerasr_amd import convert as cv
    sd = module.state_dict()
    used = 0
    for pk, fk in cv.name_map(cfg).items():
        if fk not in sd:
            continue
        a = np.asarray(w[pk], np.float32)
        if pk.endswith("fsmn.weight"):
            a = a[:, None, :]                                          # depthwise Conv1d [D, 1, k]
        assert tuple(sd[fk].shape) == a.shape, (fk, tuple(sd[fk].shape), a.shape)
        sd[fk] = torch.from_numpy(a.copy())
        used += 1
    module.load_state_dict(sd)
    return used


def export_onnx(module, args, input_names, output_names, dynamic_axes=None, opset=17) -> bytes:
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    keep = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda proto, opsets: proto     # (needs the `onnx` package; no onnxscript functions here)
    buf = io.BytesIO()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.onnx.export(module.eval(), args, buf, dynamo=False, opset_version=opset, input_names=input_names,
                              output_names=output_names, dynamic_axes=dynamic_axes, do_constant_folding=True)
    finally:
        onnx_proto_utils._add_onnxscript_fn = keep
    return buf.getvalue()


def quantize_dynamic_rewrite(blob: bytes, exclude=()):
    """What onnxruntime's `quantize_dynamic(per_channel=True, weight_type=QUInt8, op_types_to_quantize=["MatMul"])` does to an
    exported graph, restated on this repository's reader / writer: every MatMul whose second operand is a 2-D float
    initialiser becomes DynamicQuantizeLinear + MatMulInteger + Cast + Mul(activation scale) + Mul(weight scales), the weight
    stored as `<name>_quantized` (uint8 [K, N]) / `<name>_scale` / `<name>_zero_point` per output channel
    (oracle.int8.quantize_weight = the same per-channel formulas).  Nodes whose name contains an entry of `exclude` stay float
    (FunASR passes nodes_to_exclude).  Returns (bytes, {initialiser name: (q [N, K], zp, scale)})."""
    from aliparaformerasr_amd import onnx_reader as R
    from oracle.int8 import quantize_weight
    g = R.load(blob)
    ini = dict(g.initializers)
    nodes, stored = [], {}
    for n in g.nodes:
        wname = n.inputs[1] if n.op_type == "MatMul" and len(n.inputs) > 1 else None
        if wname in ini and ini[wname].ndim == 2 and ini[wname].dtype == np.float32 and not any(e in n.name for e in exclude):
            if wname not in stored:
                wq, ws, wz = quantize_weight(np.ascontiguousarray(ini[wname].T))           # [N, K] per output channel
                stored[wname] = (wq.astype(np.uint8), wz.astype(np.uint8), ws.astype(np.float32))
                ini[wname + "_quantized"] = np.ascontiguousarray(wq.T.astype(np.uint8))
                ini[wname + "_scale"], ini[wname + "_zero_point"] = ws.astype(np.float32), wz.astype(np.uint8)
            o = n.outputs[0]
            nodes.append(("DynamicQuantizeLinear", n.name + "_dql", [n.inputs[0]], [o + "_aq", o + "_as", o + "_az"], {}))
            nodes.append(("MatMulInteger", n.name + "_quant", [o + "_aq", wname + "_quantized", o + "_az", wname + "_zero_point"], [o + "_i32"], {}))
            nodes.append(("Cast", n.name + "_cast", [o + "_i32"], [o + "_f32"], {"to": 1}))
            nodes.append(("Mul", n.name + "_scales", [o + "_as", wname + "_scale"], [o + "_sc"], {}))
            nodes.append(("Mul", n.name + "_mul", [o + "_f32", o + "_sc"], [o], {}))
        else:
            nodes.append((n.op_type, n.name, list(n.inputs), list(n.outputs), dict(n.attrs)))
    for wname in stored:
        ini.pop(wname, None)
    return R.dump(nodes, ini, list(g.inputs), list(g.outputs)), stored
